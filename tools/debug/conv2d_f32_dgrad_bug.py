"""The f32-MFMA conv2d data gradient (conv2d.hip, RH_CONV2D_X6=0) is 3.6e-3 off on (2,32,32,129,61,(9,3),(2,1),(1,2),(4,2)) --
which kernel (LDS-DMA or generic: RH_CONV2D_NODMA) and which geometry feature?  One process per environment."""
import math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    import torch, torch.nn.functional as F
    from rave_amd import ops
    dev = torch.device("cuda:0")
    cases = [(2, 32, 32, 129, 61, (9, 3), (2, 1), (1, 2), (4, 2)), (2, 32, 32, 129, 61, (9, 3), (2, 1), (1, 1), (4, 1)),
             (2, 32, 32, 129, 64, (9, 3), (2, 1), (1, 2), (4, 2)), (1, 32, 32, 129, 61, (9, 3), (2, 1), (1, 2), (4, 2)),
             (2, 32, 32, 128, 61, (9, 3), (2, 1), (1, 2), (4, 2)), (2, 32, 32, 129, 61, (9, 3), (1, 1), (1, 2), (4, 2)),
             (2, 16, 32, 129, 61, (9, 3), (2, 1), (1, 2), (4, 2)), (2, 8, 8, 129, 29, (9, 3), (2, 1), (1, 2), (4, 2)),
             (2, 32, 32, 65, 125, (9, 3), (2, 1), (1, 4), (4, 4)), (2, 32, 32, 129, 61, (3, 3), (2, 1), (1, 2), (1, 2))]
    for c in cases:
        B, Ci, Co, H, W, k, s, d, p = c
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, Ci, H, W, generator=g); w = torch.randn(Co, Ci, *k, generator=g) / math.sqrt(Ci * k[0] * k[1])
        xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
        ref = F.conv2d(xd, wd, None, s, p, d)
        cot = torch.randn(ref.shape, generator=g)
        ref.backward(cot.double())
        xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        y = ops.conv2d(xg, wg, None, s, p, d)
        y.backward(cot.to(dev))
        rel = lambda a, b: float((a.detach().double().cpu() - b).norm() / b.norm())
        e = (xg.grad.detach().double().cpu() - xd.grad)
        bad = (e.abs() > 1e-4 * xd.grad.abs().max()).nonzero()
        where = ""
        if len(bad):
            hs = sorted(set(bad[:, 2].tolist())); ws = sorted(set(bad[:, 3].tolist()))
            where = f"  bad h {hs[:6]}..{hs[-3:]} ({len(hs)})  w {ws[:6]}..{ws[-3:]} ({len(ws)})  n={len(bad)}"
        print(f"{c}: fwd {rel(y, ref.detach()):.1e} dx {rel(xg.grad, xd.grad):.1e} dw {rel(wg.grad, wd.grad):.1e}{where}")
else:
    for env in ({}, {"RH_CONV2D_NODMA": "1"}):
        e = dict(os.environ, RH_CONV2D_X6="0", RH_CONV2D_SMALLM="0", **env)
        print("==", env or "default (LDS-DMA)")
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=e)
