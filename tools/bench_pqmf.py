"""PQMF kernels at the BASELINE size (32 x 65536): direct-form MFMA kernels, folded form generation 1 (pqmf_fold.hip)
and generation 2 (pqmf_fold2.hip); module-level timings (autograd call, eager) and kernel-level ones (the C ABI called
back to back: launch-to-launch time of the kernel alone, warm waveform), against the 8 B/sample HBM roofline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rave_amd import pqmf, _lib as L
dev = torch.device("cuda:0")
m = pqmf.CachedPQMF(100, 16).to(dev)
x = torch.randn(32, 1, 65536, device=dev)


def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


mb = 2 * 32 * 65536 * 4 / 1e6                     # algorithmic bytes: 8 B / sample
for fold, v2 in (("1", "1"), ("1", "0"), ("0", "0")):
    os.environ["RH_PQMF_FOLD"] = fold
    os.environ["RH_PQMF_V2"] = v2
    xa = x.clone().requires_grad_(True)
    y = m(xa); cy = torch.randn_like(y)
    ya = y.detach().clone().requires_grad_(True)
    xr = m.inverse(ya); cx = torch.randn_like(xr)
    with torch.no_grad():
        a = t(lambda: m(x)); s = t(lambda: m.inverse(y))
    ab = t(lambda: torch.autograd.grad(y, xa, cy, retain_graph=True))
    sb = t(lambda: torch.autograd.grad(xr, ya, cx, retain_graph=True))
    print("fold=%s v2=%s  module: analysis %.1f us (%.2f TB/s)  synthesis %.1f us (%.2f TB/s)  analysis-bwd %.1f us  synthesis-bwd %.1f us"
          % (fold, v2, a, mb / a, s, mb / s, ab, sb))      # MB / us = TB/s (HIP events around the autograd call)

# kernel level: the folded-form entry points themselves
os.environ["RH_PQMF_FOLD"] = "1"
ft = m._fold(m.forward_conv.weight)
tab, lpad = ft
st = torch.cuda.current_stream().cuda_stream
yb = torch.empty(32, 16, 4096, device=dev)
xo = torch.empty(32, 1, 65536, device=dev)
pad = m.forward_conv._pad
ipad = m.inverse_conv._pad
for v2 in ("1", "0"):
    os.environ["RH_PQMF_V2"] = v2
    k1 = t(lambda: L.check(L.lib.rh_pqmf_fold_k1_f32(x.data_ptr(), tab.data_ptr(), 32, 65536, 4096, lpad - pad[0], 1.0, yb.data_ptr(), st)), 50)
    k2 = t(lambda: L.check(L.lib.rh_pqmf_fold_k2_f32(yb.data_ptr(), tab.data_ptr(), 32, 4096, 65536, 496 - 16 * ipad[0] - lpad, 16.0, xo.data_ptr(), st)), 50)
    print("kernel level v2=%s: fold->matrix (analysis fwd / synthesis bwd) %.1f us = %.2f TB/s = %.1f %% of 8 TB/s;  "
          "matrix->overlap-add (synthesis fwd / analysis bwd) %.1f us = %.2f TB/s = %.1f %%"
          % (v2, k1, mb / k1, 100 * mb / k1 / 8.0, k2, mb / k2, 100 * mb / k2 / 8.0))
