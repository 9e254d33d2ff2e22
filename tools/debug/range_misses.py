"""Which tensors of a training step reach a convolution without a range slot (and therefore cost an rh_amax_f32 pass)?
   python tools/debug/range_misses.py [vae|gan] [v2|discrete|v3]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rave_amd import model as M, ops as R

phase = sys.argv[1] if len(sys.argv) > 1 else "gan"
cfg = sys.argv[2] if len(sys.argv) > 2 else "v2"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = {"v2": M.build_v2, "discrete": M.build_discrete, "v3": M.build_v3}[cfg]().to(dev).train()
m.configure_optimizers()
m.warmed_up = phase == "gan"
x = (0.3 * torch.randn(8, 2 if cfg == "v3" else 1, 65536)).clamp(-1, 1).to(dev)
for i in range(4):
    if i == 2:
        R.range_miss_log_begin()
    m.training_step(x.clone(), i)
log = R.range_miss_log_end()
cnt = collections.Counter(log)
print(f"{len(log)} amax passes over 2 steps ({phase}, {cfg}):")
for (tag, shape), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"  {n:3d} x {tag:9s} {shape}")
