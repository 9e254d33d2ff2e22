"""torch.profiler view of one v2 GAN-phase generator step: who issues the device copies / adds?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from collections import Counter
from rave_amd import model as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = M.build_v2().to(dev).train()
m.configure_optimizers()
m.warmed_up = True
x = (0.1 * torch.randn(32, 1, 65536)).to(dev)
for i in range(4):
    m.training_step(x.detach().clone(), i)
torch.cuda.synchronize()
for step in (5, 6):        # generator step (odd) and discriminator step (even)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        m.training_step(x.detach().clone(), step)
        torch.cuda.synchronize()
    cnt, tim = Counter(), Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::contiguous", "aten::clone"):
            chain, q = [], ev.cpu_parent
            while q is not None and len(chain) < 3:
                chain.append(q.name[:48])
                q = q.cpu_parent
            key = (ev.name, " <- ".join(chain), str(ev.input_shapes)[:60])
            cnt[key] += 1
            tim[key] += ev.device_time_total
    print("==== step", step, "(dis step)" if step % 2 == 0 else "(gen step)")
    for key, t in tim.most_common(22):
        print("%8.1f us x%3d  %s | %s | %s" % (t, cnt[key], *key))
