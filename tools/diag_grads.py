"""GPU diagnostic (not a test): per-parameter gradient comparison HIP vs CPU oracle on the golden
tiny model with identical cotangents injected at the hot-path outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rave_oracle as O
from conftest import rel_l2
from rave_amd import model as M, ops

dev = torch.device("cuda:0")
g = torch.load(os.path.join(ROOT, "tests/golden/v2_tiny.pt"), weights_only=False)
c = g["config"]
cfg = O.v2_config(capacity=c["capacity"], latent_size=c["latent_size"])
sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.startswith("pqmf.h")) for k, v in g["state_dict"].items()}
out = O.rave_forward(g["x"], sd, cfg, g["eps"])
torch.manual_seed(0)
c_raw = torch.randn_like(out["y_raw"]); c_mb = torch.randn_like(out["y_mb"])
for k in ("x_mb", "z_params", "z", "y_mb", "y_raw"):
    out[k].retain_grad()
torch.autograd.backward([out["y_raw"], out["y_mb"], out["reg"]], [c_raw, c_mb, torch.ones(())])

m = M.build_v2(capacity=c["capacity"], latent_size=c["latent_size"])
m.load_state_dict(g["state_dict"], strict=False)
m = m.to(dev).train()
x = g["x"].to(dev).requires_grad_(True)
zp, x_mb = m.encode(x, return_mb=True)
z, reg = m.encoder.reparametrize(zp, g["eps"].to(dev))
y_mb = m.decoder(z)
y_raw = _ = None
from rave_amd.model import _pqmf_decode
y_raw = _pqmf_decode(m.pqmf, y_mb, batch_size=z.shape[:-2], n_channels=1)
for t in (x_mb, zp, z, y_mb, y_raw):
    t.retain_grad()
print("fwd rel: x_mb %.2e zp %.2e y_mb %.2e y_raw %.2e" % (rel_l2(x_mb, out["x_mb"]), rel_l2(zp, out["z_params"]), rel_l2(y_mb, out["y_mb"]), rel_l2(y_raw, out["y_raw"])))
torch.autograd.backward([y_raw, y_mb, reg], [c_raw.to(dev), c_mb.to(dev), torch.ones((), device=dev)])
# NB: in the oracle y_mb feeds y_raw (one decoder pass); here too (single pass) -> same graph shape
print("act grads: z %.2e zp %.2e x_mb %.2e x %.2e" % (rel_l2(z.grad, out["z"].grad), rel_l2(zp.grad, out["z_params"].grad), rel_l2(x_mb.grad, out["x_mb"].grad), 0.0))
named = dict(m.named_parameters())
for k, p in named.items():
    if k in sd and sd[k].grad is not None and p.grad is not None:
        print("%-70s %.3e  |g|=%.3e" % (k, rel_l2(p.grad, sd[k].grad), float(sd[k].grad.norm())))
# pqmf synthesis backward multi-tile check vs oracle
b = O.pqmf_buffers(100, 16)
yy = torch.randn(2, 16, 2048)
yr = yy.clone().requires_grad_(True)
xr = O.pqmf_synthesis(yr, b["inverse_conv.weight"]); cc_ = torch.randn_like(xr); (xr * cc_).sum().backward()
yg = yy.to(dev).requires_grad_(True)
xg = ops.pqmf_synthesis(yg, b["inverse_conv.weight"].to(dev), (16, 16)); (xg * cc_.to(dev)).sum().backward()
print("pqmf synth multi-tile fwd %.2e bwd %.2e" % (rel_l2(xg, xr), rel_l2(yg.grad, yr.grad)))
xx = torch.randn(2, 1, 32768)
xr2 = xx.clone().requires_grad_(True)
ya = O.pqmf_analysis(xr2, b["forward_conv.weight"]); ca = torch.randn_like(ya); (ya * ca).sum().backward()
xg2 = xx.to(dev).requires_grad_(True)
yb = ops.pqmf_analysis(xg2, b["forward_conv.weight"].to(dev), (256, 256)); (yb * ca.to(dev)).sum().backward()
print("pqmf analysis multi-tile fwd %.2e bwd %.2e" % (rel_l2(yb, ya), rel_l2(xg2.grad, xr2.grad)))
