"""Which of {eager, graph} x {STFT precompute on, off} agree bit for bit after a few VAE-phase steps (capacity 16)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import rave_oracle as O
from rave_amd import model as M
dev = torch.device("cuda:0")


def run(graphed, pre, steps=8):
    os.environ["RH_STFT_PRECOMPUTE"] = "1" if pre else "0"
    torch.manual_seed(0)
    m = M.build_v2(capacity=16, latent_size=16).to(dev).train()
    m.configure_optimizers(capturable=True)
    xs = [O.synthetic_batch(2, 1, 32768, seed=50 + i).to(dev) for i in range(steps)]
    gen = torch.Generator().manual_seed(2)
    es = [torch.randn(2, 16, 16, generator=gen).to(dev) for _ in range(steps)]
    step = M.GraphedTrainingStep(m, xs[0], inject_eps=True) if graphed else None
    logs = []
    for i in range(steps):
        if graphed:
            lg = step(xs[i], i, eps=es[i])
        else:
            lg = m.training_step(xs[i].clone(), i, eps=es[i], capture_safe=True)
        logs.append({k: float(v) for k, v in lg.items() if torch.is_tensor(v)})
        m.on_train_batch_end(None, None, i)
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in m.named_parameters()}, logs


res = {}
for name, g, p in (("eager_off", 0, 0), ("graph_on_a", 1, 1), ("graph_on_b", 1, 1), ("graph_on_c", 1, 1), ("graph_on_d", 1, 1), ("graph_off", 1, 0)):
    res[name] = run(g, p)
base = res["eager_off"]
for name, (params, logs) in res.items():
    bad = [k for k in params if not torch.equal(params[k], base[0][k])]
    print(f"{name:12s} differing tensors vs eager_off: {len(bad):3d}  first-step losses {logs[0]}")
    if bad:
        k = bad[0]
        print("   e.g.", k, float((params[k] - base[0][k]).abs().max()))
