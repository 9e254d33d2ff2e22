"""Which Python lines of the step launch the small ATen kernels (fill / copy / mul / add)?  Runs eager v2 VAE-phase steps at
the benchmarked size under torch.profiler with stacks and prints, per (ATen op, innermost rave_amd / bench frame), the number
of GPU kernel launches per step.  (rocprofv3 shows ~94 such launches per step; they sit between forward and backward.)"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rave_amd import model as M  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
phase = sys.argv[1] if len(sys.argv) > 1 else "vae"
m = M.build_v2().to(dev).train()
m.configure_optimizers(capturable=True)
m.warmed_up = phase == "gan"
x = (0.1 * torch.randn(32, 1, 65536)).clamp(-1, 1).to(dev)
for i in range(3):
    m.training_step(x.detach().clone(), i, capture_safe=True)
    m.on_train_batch_end(None, None, i)
torch.cuda.synchronize()
NS = 2
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(NS):
        m.training_step(x.detach().clone(), 3 + i, capture_safe=True)
        m.on_train_batch_end(None, None, 3 + i)
    torch.cuda.synchronize()

cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or not ev.kernels:
        continue
    # innermost op only: an op whose kernels are also counted by a child would double count -> use ops with no aten child
    if any(c.name.startswith("aten::") and c.kernels for c in ev.cpu_children):
        continue
    frame = "?"
    for fr in ev.stack or ():
        if "rave_amd" in fr or "bench.py" in fr or "aten_tail" in fr:
            frame = fr.strip()
            break
    if frame == "?" and ev.stack:
        frame = "(autograd) " + " <- ".join(s.strip().split("/")[-1] for s in ev.stack[:2])
    shape = str(ev.input_shapes)[:60] if ev.input_shapes else ""
    cnt[(ev.name, frame[-110:], shape)] += len(ev.kernels)
tot = 0
for (name, frame, shape), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n / NS:6.1f}  {name:22s} {shape:60s} {frame}")
    tot += n
print(f"total ATen kernel launches per step: {tot / NS:.1f}")
