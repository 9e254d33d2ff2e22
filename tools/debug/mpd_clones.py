import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from rave_amd import model as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = M.build_v2().to(dev).train()
x = (0.1 * torch.randn(16, 1, 65536)).to(dev).requires_grad_(True)
mpd = m.discriminator.discriminators[0] if hasattr(m.discriminator, "discriminators") else m.discriminator
print(type(mpd).__name__, [type(d).__name__ for d in getattr(m.discriminator, "discriminators", [])])
for name, lossf in (("last scores only", lambda fs: sum(f[-1].mean() for f in fs)),
                    ("split last scores", lambda fs: sum(torch.split(f[-1], f[-1].shape[0] // 2, 0)[0].mean() for f in fs))):
    for d in getattr(m.discriminator, "discriminators", [m.discriminator]):
        fs = d(x)
        loss = lossf(fs)
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            loss.backward()
            torch.cuda.synchronize()
        shapes = [str(ev.input_shapes)[:50] for ev in prof.events() if ev.name == "aten::clone"]
        print(name, type(d).__name__, "clones:", len(shapes), shapes[:6])
from rave_amd import ops
d = m.discriminator.discriminators[0]
fs = d(x)
maps = [f for s in fs for f in s]
print([ (tuple(f.shape), f.is_contiguous(), ops._dense_batch_major(f)) for f in maps[:5]])
loss = ops.feature_matching(maps, [1.0 / len(maps)] * len(maps), True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    loss.backward()
    torch.cuda.synchronize()
shapes = [str(ev.input_shapes)[:50] for ev in prof.events() if ev.name == "aten::clone"]
print("fused fm on MPD: clones:", len(shapes), shapes[:8])
