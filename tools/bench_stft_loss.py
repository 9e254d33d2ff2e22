"""Kernel-level timings of the in-kernel STFT spectral distance (stft_loss.hip) through the C ABI, back to back, at the two
shapes of the v2 training step (fullband 32 x 65536, multiband 512 x 4096), against the framing + rocFFT path (module level)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rave_amd import ops, _lib as L
dev = torch.device("cuda:0")
eps = 1e-7
scales = (2048, 1024, 512, 256, 128)


def tm(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


st = torch.cuda.current_stream().cuda_stream
tot = {}
for rows, t in ((32, 65536), (512, 4096)):
    x = torch.randn(rows, t, device=dev); y = torch.randn(rows, t, device=dev)
    dx = torch.empty_like(x); dy = torch.empty_like(y)
    g = torch.ones(1, device=dev)
    for n in scales:
        w = torch.hann_window(n, dtype=torch.float64); w = (w / w.pow(2).sum().sqrt()).float().to(dev)
        tw = ops._twiddle(n, dev)
        nb = L.lib.rh_stft_loss_workspace_bytes(n, t, rows)
        ws = torch.empty(nb // 4, device=dev); sums = torch.empty(3, device=dev)
        f = tm(lambda: L.check(L.lib.rh_stft_loss_fwd_f32(L.ptr(x), L.ptr(y), L.ptr(w), L.ptr(tw), rows, t, n, eps, L.ptr(sums), L.ptr(ws), nb, st)))
        b = tm(lambda: L.check(L.lib.rh_stft_loss_bwd_f32(L.ptr(x), L.ptr(y), L.ptr(w), L.ptr(tw), rows, t, n, eps, L.ptr(sums), L.ptr(g), L.ptr(dx), L.ptr(dy), 1, st)))
        b1 = tm(lambda: L.check(L.lib.rh_stft_loss_bwd_f32(L.ptr(x), L.ptr(y), L.ptr(w), L.ptr(tw), rows, t, n, eps, L.ptr(sums), L.ptr(g), None, L.ptr(dy), 1, st)))
        nf = t // (n // 4) + 1
        print("rows %3d t %5d n_fft %4d: forward (+finalize) %6.1f us   backward dx+dy %6.1f us   dy only %6.1f us   [%d frame pairs, signals %.1f MB]"
              % (rows, t, n, f, b, b1, rows * nf, 2 * rows * t * 4 / 1e6))
        tot[(rows, "f")] = tot.get((rows, "f"), 0) + f; tot[(rows, "b")] = tot.get((rows, "b"), 0) + b
for k, v in tot.items():
    print("sum over scales rows %d %s: %.1f us" % (k[0], "forward" if k[1] == "f" else "backward", v))
