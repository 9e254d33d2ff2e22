import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rave_oracle as O
from conftest import rel_l2
from rave_amd import model as M
from rave_amd.model import _pqmf_decode

dev = torch.device("cuda:0")
g = torch.load(os.path.join(ROOT, "tests/golden/v2_tiny.pt"), weights_only=False)
c = g["config"]
cfg = O.v2_config(capacity=c["capacity"], latent_size=c["latent_size"])
KEYS = ["encoder.encoder.net.0.weight_v", "encoder.encoder.net.5.weight_v", "encoder.encoder.net.21.weight_v",
        "decoder.net.0.weight_v", "decoder.net.11.weight_v", "decoder.net.21.weight_v"]

def cpu(c_raw, c_mb):
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.startswith("pqmf.h")) for k, v in g["state_dict"].items()}
    out = O.rave_forward(g["x"], sd, cfg, g["eps"])
    torch.autograd.backward([out["y_raw"], out["y_mb"], out["reg"]], [c_raw, c_mb, torch.ones(())])
    return sd

def gpu(c_raw, c_mb, twice, xgrad):
    m = M.build_v2(capacity=c["capacity"], latent_size=c["latent_size"])
    m.load_state_dict(g["state_dict"], strict=False)
    m = m.to(dev).train()
    x = g["x"].to(dev)
    if xgrad: x.requires_grad_(True)
    zp, x_mb = m.encode(x, return_mb=True)
    z, reg = m.encoder.reparametrize(zp, g["eps"].to(dev))
    y_mb = m.decoder(z)
    y_raw = m.decode(z) if twice else _pqmf_decode(m.pqmf, y_mb, batch_size=z.shape[:-2], n_channels=1)
    torch.autograd.backward([y_raw, y_mb, reg], [c_raw.to(dev), c_mb.to(dev), torch.ones((), device=dev)])
    return dict(m.named_parameters())

# loss cotangents from the oracle
sd0 = {k: v.clone().requires_grad_(v.is_floating_point() and not k.startswith("pqmf.h")) for k, v in g["state_dict"].items()}
x0 = g["x"].clone().requires_grad_(True)
loss, _, parts, out = O.generator_losses(x0, sd0, cfg, g["eps"], warmed_up=False)
for k in ("y_raw", "y_mb"): out[k].retain_grad()
loss.backward()
L_raw, L_mb = out["y_raw"].grad.clone(), out["y_mb"].grad.clone()
torch.manual_seed(0)
W_raw, W_mb = torch.randn_like(L_raw), torch.randn_like(L_mb)
for name, (cr, cm) in {"white": (W_raw, W_mb), "loss": (L_raw, L_mb), "loss_raw_only": (L_raw, torch.zeros_like(L_mb)), "loss_mb_only": (torch.zeros_like(L_raw), L_mb)}.items():
    ref = cpu(cr, cm)
    for twice in (False, True):
        for xgrad in (True, False):
            got = gpu(cr, cm, twice, xgrad)
            print(name, "twice", twice, "xgrad", xgrad, " ".join("%.1e" % rel_l2(got[k].grad, ref[k].grad) for k in KEYS))
    if name == "loss":
        print("   cpu-inject vs golden", " ".join("%.1e" % rel_l2(ref[k].grad, g["vae"]["grads"][k]) for k in KEYS if k in g["vae"]["grads"]))
