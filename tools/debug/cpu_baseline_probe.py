import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rave_oracle as O
print("cpus", os.cpu_count(), flush=True)
cfg = O.v2_config(); sd = O.init_state_dict(cfg, seed=0, with_discriminator=False)
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(("encoder.", "decoder."))}
full = dict(sd); full.update(leaves)
for b, n in ((8, 256), (8, 128), (8, 64), (8, 32), (1, 16)):
    torch.set_num_threads(n)
    x = O.synthetic_batch(b, 1, 65536); eps = torch.randn(b, 128, 32)
    ts = []
    for i in range(3):
        t = time.perf_counter()
        for v in leaves.values(): v.grad = None
        loss = O.generator_losses(x.clone().requires_grad_(True), full, cfg, eps, warmed_up=False)[0]
        loss.backward()
        ts.append(time.perf_counter() - t)
    print(b, n, ["%.2f" % t for t in ts], "samples/s %.3g" % (b * 65536 / min(ts)), flush=True)
