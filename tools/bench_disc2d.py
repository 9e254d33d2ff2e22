"""GPU benchmark of the general-Conv2d discriminators at BASELINE sizes (config 4: spectral Encodec
discriminator; config 5: descript MPD + MRD, stereo): forward + backward with per-launch HIP-event timing.
    N=<waveforms> WHICH=encodec|descript python tools/bench_disc2d.py"""
import os, sys
from collections import OrderedDict
from functools import partial
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rave_amd import ops, discriminator as D, descript_discriminator as DD

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 8))
which = os.environ.get("WHICH", "encodec")
torch.manual_seed(0)
if which == "encodec":
    model = D.MultiScaleSpectralDiscriminator([4096, 2048, 1024, 512, 256], partial(D.EncodecConvNet, capacity=32), n_channels=1)
    x = torch.randn(N, 1, 65536, device=dev) * 0.1
elif which == "v2":
    from rave_amd import model as M
    model = M.build_v2().discriminator
    x = torch.randn(N, 1, 65536, device=dev) * 0.1
else:
    model = DD.DescriptDiscriminator(n_channels=2)
    x = torch.randn(N, 2, 65536, device=dev) * 0.1
model.to(dev)
x.requires_grad_(True)


def step():
    feats = model(x)
    loss = sum(f.pow(2).mean() for net in feats for f in net)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 3
for _ in range(reps):
    step()
e1.record()
torch.cuda.synchronize()
print(f"{which}: {N} waveforms x 65536, fwd+bwd {e0.elapsed_time(e1) / reps:.2f} ms")

# per-launch timing: wrap _launch2 to also record the geometry
recs = []
orig, orig1 = ops._launch2, ops._launch


def spy(kind, d, fn):
    key = (kind, d.c_in, d.c_out, d.h_in, d.w_in, d.kh, d.kw, d.sh, d.sw, d.dh, d.dw)
    recs.append(key)
    return orig(kind, d, fn)


def spy1(kind, d, fn, has_bias=False, has_add=False):      # (k,1) convs routed to the 1-D kernels: H = l_in, W = inner
    key = ("(1d)", d.c_in, d.c_out, d.l_in, d.inner, d.kernel, 1, d.stride, 1, d.dilation, 1)
    recs.append(key)
    return orig1(kind, d, fn, has_bias, has_add)


ops._launch2, ops._launch = spy, spy1
# per-launch figures are only meaningful one kernel at a time: the weight-gradient side stream (which overlaps the
# data-gradient chain in the timed step above) is switched off for the instrumented step
os.environ["RH_BWD_SIDE_STREAM"] = "0"
ops.profile_begin()
step()
rec = ops.profile_end()
os.environ.pop("RH_BWD_SIDE_STREAM", None)
ops._launch2, ops._launch = orig, orig1
agg = OrderedDict()
for key, (kind, fl, by, kms, cms) in zip(recs, rec):     # kms = the main kernel's own duration; cms = the whole call
    ms = kms if kms is not None else cms
    if key[0] == "(1d)":       # the recorded kind carries the kernel family tag ([x6] / [f32])
        key = (kind + "(1d)",) + key[1:]
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += fl; a[2] += ms
tot = {}
print("%-17s %5s %5s %6s %5s %6s %4s %3s | %3s %9s %9s %7s" % ("kind", "cin", "cout", "H", "W", "k", "s", "d", "n", "GFLOP", "ms", "TF/s"))
for key, (n, fl, ms) in agg.items():
    kind, ci, co, h, w, kh, kw, sh, sw, dh, dw = key
    print("%-17s %5d %5d %6d %5d %6s %4s %3s | %3d %9.2f %9.3f %7.2f" % (kind, ci, co, h, w, f"{kh}x{kw}", f"{sh}{sw}", f"{dh}{dw}", n, fl / 1e9, ms, fl / ms / 1e9))
    t = tot.setdefault(kind, [0.0, 0.0]); t[0] += fl; t[1] += ms
for kind, (fl, ms) in tot.items():
    print(f"TOTAL {kind}: {fl / 1e12:.3f} TFLOP in {ms:.2f} ms = {fl / ms / 1e9:.2f} TF/s")
