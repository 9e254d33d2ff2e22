"""Builder-side diagnosis: in-kernel STFT loss gradient error vs f64 per scale with the pair equaliser on (default) / off
(RH_STFT_EQUALISE=0, read once per process -> this script re-executes itself), quiet smooth y against a loud x."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for eq in ("1", "0"):
        env = dict(os.environ, RH_STFT_EQUALISE=eq)
        print(f"==== RH_STFT_EQUALISE={eq}", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env)
    sys.exit(0)
import torch
from rave_amd import ops
dev = torch.device("cuda:0")


def win(n):
    w = torch.hann_window(n, dtype=torch.float64)
    return (w / w.pow(2).sum().sqrt())


def ref(x, y, n, dt, dvc):
    x = x.detach().to(dvc).to(dt).clone().requires_grad_(True); y = y.detach().to(dvc).to(dt).clone().requires_grad_(True)
    w = win(n).to(dvc).to(dt)
    sx = torch.stft(x, n, n // 4, n, w, center=True, pad_mode="reflect", return_complex=True).abs()
    sy = torch.stft(y, n, n // 4, n, w, center=True, pad_mode="reflect", return_complex=True).abs()
    d = ((sx - sy) ** 2).mean() / (sx ** 2).mean() + (torch.log(sx + 1e-7) - torch.log(sy + 1e-7)).abs().mean()
    d.backward()
    return x.grad, y.grad


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())


for rows, t in ((64, 4096), (4, 65536), (8, 16384)):
    g = torch.Generator().manual_seed(rows + t)
    loud = (0.1 * torch.randn(rows, t, generator=g) + 0.2 * torch.sin(torch.arange(t) * 0.0313)[None]).clamp(-1, 1)
    q = torch.randn(rows, t + 32, generator=g)
    q = torch.nn.functional.avg_pool1d(q[:, None], 33, 1)[:, 0] * 0.02
    x, y = loud.to(dev), q.to(dev)
    for n in (2048, 1024, 512, 256, 128):
        gx64, gy64 = ref(x, y, n, torch.float64, "cpu")
        gxt, gyt = ref(x, y, n, torch.float32, dev)
        xx = x.detach().clone().requires_grad_(True); yy = y.detach().clone().requires_grad_(True)
        ops.multiscale_stft_distance(xx, yy, [win(n).float().to(dev)], [n], 1e-7).backward()
        e = (yy.grad.double().cpu() - gy64).pow(2).sum(0)
        k = max(1, int(0.05 * t))
        print(f"rows {rows:3d} t {t:6d} n {n:5d}: kernel dy {rel(yy.grad, gy64):.2e} dx {rel(xx.grad, gx64):.2e} | torch f32 dy {rel(gyt, gy64):.2e} dx {rel(gxt, gx64):.2e}"
              f" | share of dy error in first/last 5% of samples {float(e[:k].sum() / e.sum()):.2f}/{float(e[-k:].sum() / e.sum()):.2f}")
