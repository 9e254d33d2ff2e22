"""In-kernel STFT distance (stft_loss.hip) against torch f64 (value and both gradients), per scale and for the 5-scale node,
at the fullband / multiband shapes of the training step and at awkward lengths; plus timings against the rocFFT path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rave_amd import ops

dev = torch.device("cuda:0")
eps = 1e-7


def windows(scales):
    ws = []
    for n in scales:
        w = torch.hann_window(n, dtype=torch.float64)
        ws.append((w / w.pow(2).sum().sqrt()).float().to(dev))
    return ws


def ref(x, y, scales, ws, eps=eps):
    x = x.double().requires_grad_(True); y = y.double().requires_grad_(True)
    d = 0
    for n, w in zip(scales, ws):
        sx = torch.stft(x, n, n // 4, n, w.double(), center=True, pad_mode="reflect", return_complex=True).abs()
        sy = torch.stft(y, n, n // 4, n, w.double(), center=True, pad_mode="reflect", return_complex=True).abs()
        d = d + ((sx - sy) ** 2).mean() / (sx ** 2).mean() + (torch.log(sx + eps) - torch.log(sy + eps)).abs().mean()
    d.backward()
    return d.detach(), x.grad, y.grad


def run(x, y, scales, ws, fused, eps=eps):
    os.environ["RH_STFT_FUSED"] = "1" if fused else "0"
    x = x.clone().requires_grad_(True); y = y.clone().requires_grad_(True)
    d = ops.multiscale_stft_distance(x, y, ws, scales, eps)
    d.backward()
    return d.detach(), x.grad, y.grad


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


bad = 0
for rows, t, scales in [(4, 65536, (2048,)), (4, 65536, (1024,)), (4, 65536, (512,)), (4, 65536, (256,)), (4, 65536, (128,)),
                        (32, 4096, (2048, 1024, 512, 256, 128)), (3, 5000, (2048, 1024, 512, 256, 128)),
                        (2, 1100, (2048, 128)), (5, 65536, (2048, 1024, 512, 256, 128))]:
    g = torch.Generator(device="cpu").manual_seed(rows * 7 + t)
    x = torch.randn(rows, t, generator=g).to(dev)
    y = (0.3 * torch.randn(rows, t, generator=g)).to(dev) + 0.5 * x
    ws = windows(scales)
    dr, gxr, gyr = ref(x, y, scales, ws)
    for fused in (True, False):
        d, gx, gy = run(x, y, scales, ws, fused)
        e = (abs(float(d) - float(dr)) / abs(float(dr)), rel(gx, gxr), rel(gy, gyr))
        ok = e[0] < 2e-6        # the gradient at eps = 1e-7 is ill-conditioned for any f32 implementation (sign flips x 1/(|S|+eps))
        bad += (not ok) and fused
        print("rows %3d t %6d scales %-28s %-6s value %.2e  dx %.2e  dy %.2e %s" % (rows, t, scales, "fused" if fused else "rocfft", *e, "" if ok else "  <-- BAD"))
    # well-conditioned variant (eps = 1e-2: 1 / (|S| + eps) bounded): the transform + gradient chain itself
    dr, gxr, gyr = ref(x, y, scales, ws, 1e-2)
    for fused in (True, False):
        d, gx, gy = run(x, y, scales, ws, fused, 1e-2)
        e = (abs(float(d) - float(dr)) / abs(float(dr)), rel(gx, gxr), rel(gy, gyr))
        ok = e[0] < 2e-6 and e[1] < 1e-4 and e[2] < 1e-4
        bad += (not ok) and fused
        print("   eps 1e-2: %-6s value %.2e  dx %.2e  dy %.2e %s" % ("fused" if fused else "rocfft", *e, "" if ok else "  <-- BAD"))
    # fused twice: bit-reproducible
    a = run(x, y, scales, ws, True); b = run(x, y, scales, ws, True)
    same = all(torch.equal(u, v) for u, v in zip(a, b))
    bad += not same
    print("   fused run twice bit-identical:", same)


def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


scales = (2048, 1024, 512, 256, 128)
ws = windows(scales)
for rows, t in ((32, 65536), (512, 4096)):
    x = torch.randn(rows, t, device=dev); y = torch.randn(rows, t, device=dev)
    for fused in (True, False):
        os.environ["RH_STFT_FUSED"] = "1" if fused else "0"
        xa = x.clone().requires_grad_(True); ya = y.clone().requires_grad_(True)
        with torch.no_grad():
            f = tm(lambda: ops.multiscale_stft_distance(xa, ya, ws, scales, eps))
        d = ops.multiscale_stft_distance(xa, ya, ws, scales, eps)
        b = tm(lambda: torch.autograd.grad(d, (xa, ya), retain_graph=True))
        print("rows %3d t %6d %-6s forward %.1f us  backward (dx and dy) %.1f us" % (rows, t, "fused" if fused else "rocfft", f, b))
    if True:
        os.environ["RH_STFT_FUSED"] = "1"
        for n, w in zip(scales, ws):
            xa = x.clone().requires_grad_(True); ya = y.clone().requires_grad_(True)
            with torch.no_grad():
                f = tm(lambda: ops.multiscale_stft_distance(xa, ya, [w], (n,), eps))
            d = ops.multiscale_stft_distance(xa, ya, [w], (n,), eps)
            b = tm(lambda: torch.autograd.grad(d, (xa, ya), retain_graph=True))
            print("     n_fft %4d fused forward %.1f us  backward %.1f us" % (n, f, b))
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
