import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import rave_oracle as O
from rave_amd import model as M
dev = torch.device("cuda:0")
NS = int(os.environ.get("NS", 2))
def run(graphed):
    torch.manual_seed(0)
    m = M.build_v2(capacity=16, latent_size=16).to(dev).train()
    m.configure_optimizers(capturable=True)
    xs = [O.synthetic_batch(2, 1, 32768, seed=50 + i).to(dev) for i in range(NS)]
    gen = torch.Generator().manual_seed(2)
    es = [torch.randn(2, 16, 16, generator=gen).to(dev) for _ in range(NS)]
    step = M.GraphedTrainingStep(m, xs[0], inject_eps=True) if graphed else None
    grads = None
    for i in range(NS):
        if graphed: step(xs[i], i, eps=es[i])
        else: m.training_step(xs[i].clone(), i, eps=es[i], capture_safe=True)
        m.on_train_batch_end(None, None, i)
        if i == 0:
            torch.cuda.synchronize()
            grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in m.named_parameters()}, grads, (step.logged if graphed else m.logged)
pe, ge, le = run(False); pg, gg, lg = run(True)
print("losses", {k: (float(le[k]), float(lg[k])) for k in le if torch.is_tensor(le[k])})
bad = [(k, float((ge[k] - gg[k]).abs().max()), float(ge[k].abs().max())) for k in ge if not torch.equal(ge[k], gg[k])]
print("grad mismatches after step 0:", len(bad), "of", len(ge)); print(bad[:8])
badp = [(k, float((pe[k] - pg[k]).abs().max())) for k in pe if not torch.equal(pe[k], pg[k])]
print("param mismatches:", len(badp), "of", len(pe)); print(badp[:8])
