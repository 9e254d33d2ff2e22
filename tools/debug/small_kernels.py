"""Builder-side diagnosis: every GPU kernel of one eager v2 VAE-phase step (batch 32 x 65536) that is NOT a librave_hip kernel --
fills, copies, ATen elementwise -- with the chain of CPU ops that launched it (torch.profiler: op -> parent ops up to the
autograd node), counted per step."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rave_amd import model as M  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = M.build_v2().to(dev).train()
m.configure_optimizers(capturable=True)
x = (0.1 * torch.randn(32, 1, 65536)).clamp(-1, 1).to(dev)
for i in range(3):
    m.training_step(x.detach().clone(), i, capture_safe=True); m.on_train_batch_end(None, None, i)
torch.cuda.synchronize()
NS = 2
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(NS):
        m.training_step(x.detach().clone(), 3 + i, capture_safe=True); m.on_train_batch_end(None, None, 3 + i)
    torch.cuda.synchronize()
cnt = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if not ev.kernels:
        continue
    if any(c.kernels for c in ev.cpu_children):
        continue
    for k in ev.kernels:
        kn = k.name
        if "rocclr" not in kn and "at::native" not in kn and "Memcpy" not in kn and "Memset" not in kn:
            continue
        chain, p = [ev.name], ev.cpu_parent
        while p is not None and len(chain) < 5:
            chain.append(p.name); p = p.cpu_parent
        key = (kn[:48], str(ev.input_shapes)[:50], " <- ".join(chain)[:150])
        cnt[key] += 1; dur[key] += k.duration
tot = 0
for key, n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n / NS:5.1f} x {dur[key] / n:6.1f} us  {key[0]:48s} {key[1]:50s} {key[2]}")
    tot += n
print("non-librave kernels per step:", tot / NS)
