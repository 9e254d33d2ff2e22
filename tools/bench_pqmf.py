"""PQMF kernels at the BASELINE size (32 x 65536): folded fast form vs direct-form MFMA kernels, all four transforms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rave_amd import pqmf
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
for mode in ("1", "0"):
    os.environ["RH_PQMF_FOLD"] = mode
    xa = x.clone().requires_grad_(True)
    y = m(xa); cy = torch.randn_like(y)
    ya = y.detach().clone().requires_grad_(True)
    xr = m.inverse(ya); cx = torch.randn_like(xr)
    with torch.no_grad():
        a = t(lambda: m(x)); s = t(lambda: m.inverse(y))
    ab = t(lambda: torch.autograd.grad(y, xa, cy, retain_graph=True))
    sb = t(lambda: torch.autograd.grad(xr, ya, cx, retain_graph=True))
    mb = 2 * 32 * 65536 * 4 / 1e6                     # algorithmic bytes: 8 B / sample
    print("fold=%s  analysis %.1f us (%.2f TB/s)  synthesis %.1f us (%.2f TB/s)  analysis-bwd %.1f us  synthesis-bwd %.1f us"
          % (mode, a, mb / a, s, mb / s, ab, sb))      # MB / us = TB/s (HIP events around the autograd call)
