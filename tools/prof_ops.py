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
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=50, max_src_column_width=110))
