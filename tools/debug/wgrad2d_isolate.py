"""wgrad2d_x6 (generalised: packed (tap, channel) columns, stride-2 parity planes) one geometry per PROCESS, each under its own
timeout, against the f32-MFMA weight gradient (RH_WGRAD2D_X6=0) on the same operands.  Ordered from the class proven in
profiles (C = 32, stride 1 along W) to the new ones."""
import math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [
    (2, 32, 32, 257, 61, (9, 3), (2, 1), (1, 1), (4, 1)),
    (2, 32, 32, 33, 253, (3, 3), (1, 1), (1, 1), (1, 1)),
    (2, 2, 32, 257, 61, (9, 3), (1, 1), (1, 1), (4, 1)),          # Encodec first layer
    (2, 4, 32, 129, 102, (3, 9), (1, 1), (1, 1), (1, 4)),         # MRD first layer (stereo)
    (2, 32, 32, 129, 102, (3, 9), (1, 2), (1, 1), (1, 4)),        # MRD stride 2 along W
    (2, 32, 1, 129, 51, (3, 3), (1, 1), (1, 1), (1, 1)),          # scoring conv
    (2, 6, 7, 19, 23, (4, 3), (3, 2), (2, 2), (3, 2)),
    (4, 2, 32, 2049, 61, (9, 3), (1, 1), (1, 1), (4, 1)),         # full-size first layer
    (4, 32, 32, 129, 256, (3, 9), (1, 2), (1, 1), (1, 4)),
    (4, 32, 32, 513, 16, (3, 9), (1, 2), (1, 1), (1, 4)),         # narrow MRD bands: 8 / 16 output columns per row
    (4, 32, 32, 513, 8, (3, 3), (1, 1), (1, 1), (1, 1)),
    (4, 32, 32, 257, 32, (3, 9), (1, 2), (1, 1), (1, 4)),
    (2, 16, 16, 9, 1, (3, 1), (1, 1), (1, 1), (1, 0)),
]
if len(sys.argv) > 2 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    import torch
    from rave_amd import ops
    dev = torch.device("cuda:0")
    B, Ci, Co, H, W, k, s, d, p = CASES[int(sys.argv[2])]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Ci, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, *k, generator=g) / math.sqrt(Ci * k[0] * k[1])).to(dev)
    res = {}
    cot = None
    for mode in ("0", "1"):
        os.environ["RH_WGRAD2D_X6"] = mode
        wg = w.clone().requires_grad_(True)
        y = ops.conv2d(x, wg, None, s, p, d)
        if cot is None:
            cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        y.backward(cot)
        torch.cuda.synchronize()
        res[mode] = wg.grad.detach().double().cpu()
    e = float((res["1"] - res["0"]).norm() / res["0"].norm())
    print(f"case {sys.argv[2]} {CASES[int(sys.argv[2])]}: x6 vs f32 rel-L2 {e:.2e}", flush=True)
else:
    for i in range(len(CASES)):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", str(i)], timeout=90, capture_output=True, text=True)
            print((r.stdout.strip().splitlines() or ["(no output)"])[-1], "" if r.returncode == 0 else f"rc={r.returncode} {r.stderr[-300:]}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"case {i}: TIMEOUT -- stopping", flush=True)
            break
