import os, sys, time
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
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, (time.perf_counter() - t0) * 1e6 / n
rows, t = 512, 4096
x = torch.randn(rows, t, device=dev); y = torch.randn(rows, t, device=dev)
for scales in [(2048,), (2048, 1024), (1024, 2048), (2048, 128), (128, 2048), (512, 256), (2048, 1024, 512, 256, 128), (128, 256, 512, 1024, 2048)]:
    ws = windows(scales)
    xa = x.clone().requires_grad_(True); ya = y.clone().requires_grad_(True)
    d = ops.multiscale_stft_distance(xa, ya, ws, scales, eps)
    print(scales, "bwd gpu %.1f us host %.1f us" % tm(lambda: torch.autograd.grad(d, (xa, ya), retain_graph=True)))
    print(scales, "bwd dy only gpu %.1f us host %.1f us" % tm(lambda: torch.autograd.grad(d, (ya,), retain_graph=True)))
