"""Per-parameter gradient error of the full-size descript discriminator hinge step: HIP vs fp64 oracle, fp32 oracle vs fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import rave_oracle as O
from rave_amd import descript_discriminator as DD
dev = torch.device("cuda:0")
periods, ffts = [int(p) for p in os.environ.get("PERIODS", "2,3,5,7,11").split(",")], [2048, 1024, 512]
T = int(os.environ.get("T", 65536))
model = DD.DescriptDiscriminator(periods=periods, fft_sizes=ffts, n_channels=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith(".window")}
sd = O.seeded_state_dict(shapes, 11)
model.load_state_dict(sd, strict=False); model.to(dev).train()
xy = torch.cat([O.synthetic_batch(1, 2, T, seed=45), O.synthetic_batch(1, 2, T, seed=46) * 0.7], 0)
def hinge(feats):
    loss = 0.0
    for net in feats:
        real, fake = net[-1][: net[-1].shape[0] // 2], net[-1][net[-1].shape[0] // 2:]
        loss = loss + torch.relu(1 - real).mean() + torch.relu(1 + fake).mean()
    return loss
l32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
f32 = O.descript_discriminator(xy, {"d." + k: v for k, v in l32.items()}, "d", periods, ffts)
for n in f32:
    for f in n: f.retain_grad()
hinge(f32).backward()
l64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
f64 = O.descript_discriminator(xy.double(), {"d." + k: v for k, v in l64.items()}, "d", periods, ffts)
for n in f64:
    for f in n: f.retain_grad()
hinge(f64).backward()
got = model(xy.to(dev))
for n in got:
    for f in n: f.retain_grad()
hinge(got).backward()
print("feature-map gradients (rel err vs fp64): net layer hip cpu32 shape")
for i, (n, n32, n64) in enumerate(zip(got, f32, f64)):
    for j, (f, a, b) in enumerate(zip(n, n32, n64)):
        if b.grad is None: continue
        nb = float(b.grad.norm())
        print(i, j, "%.2e %.2e" % (float((f.grad.double().cpu() - b.grad).norm()) / nb, float((a.grad.double() - b.grad).norm()) / nb), tuple(f.shape))
print("parameter gradients: hip cpu32 norm name")
for k, p in model.named_parameters():
    w = l64[k].grad
    if w is None: continue
    nb = float(w.norm()) + 1e-30
    e = float((p.grad.double().cpu() - w).norm()) / nb
    r = float((l32[k].grad.double() - w).norm()) / nb
    if e > 2e-4: print("%.2e %.2e %.2e %s" % (e, r, nb, k))
