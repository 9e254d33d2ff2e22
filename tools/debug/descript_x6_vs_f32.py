"""Full-size descript MPD (one period): HIP with RH_CONV_X6=1 vs =0 -- forward features, gate flips, feature gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import rave_oracle as O
from rave_amd import descript_discriminator as DD
dev = torch.device("cuda:0")
periods, ffts = [11], [2048]
model = DD.DescriptDiscriminator(periods=periods, fft_sizes=ffts, n_channels=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith(".window")}
sd = O.seeded_state_dict(shapes, 11)
model.load_state_dict(sd, strict=False); model.to(dev).train()
xy = torch.cat([O.synthetic_batch(1, 2, 65536, seed=45), O.synthetic_batch(1, 2, 65536, seed=46) * 0.7], 0).to(dev)
gen = torch.Generator().manual_seed(5)
res = {}
for mode in ("1", "0"):
    os.environ["RH_CONV_X6"] = mode
    model.zero_grad()
    feats = model(xy)
    net = feats[0]
    for f in net: f.retain_grad()
    if "cot" not in res:
        res["cot"] = torch.randn(net[-1].shape, generator=gen).to(dev) / net[-1].numel()
    net[-1].backward(res["cot"])
    res[mode] = ([f.detach().clone() for f in net], [f.grad.clone() for f in net])
for i, (a, b, ga, gb) in enumerate(zip(res["1"][0], res["0"][0], res["1"][1], res["0"][1])):
    flips = int(((a > 0) != (b > 0)).sum())
    print(i, tuple(a.shape), "fwd rel %.2e  sign flips %d  grad rel %.2e" % (float((a - b).norm() / b.norm()), flips, float((ga - gb).norm() / gb.norm())))
    if flips:
        idx = ((a > 0) != (b > 0)).nonzero()[:5]
        for t in idx:
            t = tuple(int(v) for v in t)
            print("    flip at", t, float(a[t]), float(b[t]), "grad", float(ga[t]), float(gb[t]), "gnorm", float(gb.norm()))
