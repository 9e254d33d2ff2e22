"""torch.profiler view of one v2 VAE-phase step: which ATen ops (copies, fills, elementwise) surround the HIP kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from rave_amd import model as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = M.build_v2().to(dev).train()
m.configure_optimizers()
x = (0.1 * torch.randn(int(os.environ.get("B", 32)), 1, 65536)).to(dev)
for i in range(3):
    m.training_step(x.detach().clone(), i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.training_step(x.detach().clone(), 3)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
for ev in prof.key_averages(group_by_stack_n=10):
    if ev.key in ("aten::clone", "aten::copy_", "aten::contiguous", "aten::fill_", "aten::add", "aten::mul"):
        print(f"{ev.key:18s} n={ev.count:3d} cuda={ev.device_time_total:9.1f}us  | " + " <- ".join(x.strip()[-70:] for x in ev.stack[:6]))

from collections import Counter
cnt = Counter()
for ev in prof.events():
    if ev.name in ("aten::clone", "aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::div", "aten::neg", "aten::sum", "aten::zeros", "aten::zeros_like", "aten::ones_like"):
        chain, q = [], ev.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name[:60])
            q = q.cpu_parent
        cnt[(ev.name, " <- ".join(chain))] += 1
for (k, c), n in cnt.most_common(60):
    print(f"PARENT {k:12s} x{n:3d}  {c}")
