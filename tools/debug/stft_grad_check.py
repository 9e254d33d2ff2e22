"""Builder-side diagnosis (not product, not a test): the gradient of the multi-scale spectral distance w.r.t. y on the REAL
hot-path outputs of a random-init v2 model (the fixture of tests/test_gpu_step_separation.py): in-kernel STFT loss (shipped),
the framing + rocFFT form (RH_STFT_FUSED=0), torch.stft autograd on the GPU in f32, CPU f32, CPU f64 -- relative L2 of each
against CPU f64, per signal, and where the difference sits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
import rave_oracle as O  # noqa: E402
from rave_amd import model as M  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm())


def main():
    dev = torch.device("cuda:0")
    batch = int(os.environ.get("BATCH", "4"))
    cfg = O.v2_config()
    sd = O.init_state_dict(cfg, seed=0, with_discriminator=False)
    x = O.synthetic_batch(batch, 1, 65536, seed=300)
    eps = torch.randn(batch, 128, 32, generator=torch.Generator().manual_seed(1000))
    m = M.build_v2()
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).train()
    with torch.no_grad():
        m.prepare_weights()
        zp, x_mb = m.encode(x.to(dev), return_mb=True)
        z = m.encoder.reparametrize(zp, eps.to(dev))[0]
        y_mb = m.decoder(z)[..., :4096]
        y_raw = m.decode(z)[..., :65536]
        m.release_weights()
    for name, xs, ys, mod in (("fullband", x.to(dev), y_raw, m.audio_distance), ("multiband", x_mb, y_mb, m.multiband_audio_distance)):
        print(f"== {name}: x {tuple(xs.shape)} rms {float(xs.pow(2).mean().sqrt()):.3e}, y rms {float(ys.pow(2).mean().sqrt()):.3e}")
        grads, vals = {}, {}
        for tag, env in (("fused", "1"), ("rocfft", "0")):
            os.environ["RH_STFT_FUSED"] = env
            yy = ys.detach().clone().requires_grad_(True)
            xx = xs.detach().clone().requires_grad_(True)
            d = sum(mod(xx, yy).values())
            d.backward()
            grads[tag], vals[tag] = yy.grad.clone(), float(d)
            grads[tag + "_dx"] = xx.grad.clone()
        os.environ.pop("RH_STFT_FUSED", None)
        for tag, dvc, dt in (("torch_gpu_f32", dev, torch.float32), ("cpu_f32", "cpu", torch.float32), ("cpu_f64", "cpu", torch.float64)):
            yy = ys.detach().to(dvc).to(dt).requires_grad_(True)
            xx = xs.detach().to(dvc).to(dt).requires_grad_(True)
            d = O.audio_distance_v1(xx, yy, cfg)
            d.backward()
            grads[tag], vals[tag] = yy.grad.clone(), float(d)
            grads[tag + "_dx"] = xx.grad.clone()
        ref = grads["cpu_f64"]
        print("   values:", {k: f"{v:.7f}" for k, v in vals.items()})
        for k in ("fused", "rocfft", "torch_gpu_f32", "cpu_f32"):
            print(f"   d/dy {k:14s} vs cpu_f64: rel-L2 {rel(grads[k], ref):.3e}   d/dx: {rel(grads[k + '_dx'], grads['cpu_f64_dx']):.3e}")
        print(f"   d/dy fused vs rocfft: {rel(grads['fused'], grads['rocfft']):.3e}")
        # where does the fused difference sit?
        dlt = (grads["fused"].double().cpu() - ref.double().cpu()).reshape(ref.shape[0] * (ref.shape[1] if ref.dim() == 3 else 1), -1)
        e_rows = dlt.pow(2).sum(1)
        e_pos = dlt.pow(2).sum(0)
        top = e_pos.topk(5)
        print(f"   error energy by row (share): {[round(float(v), 3) for v in (e_rows / e_rows.sum())[:8]]}")
        print(f"   top-5 positions carry {float(top.values.sum() / e_pos.sum()):.3f} of the error energy: {top.indices.tolist()}")
        n = dlt.shape[1]
        edge = int(0.02 * n)
        print(f"   share of the error in the first / last 2 % of the samples: {float(e_pos[:edge].sum() / e_pos.sum()):.3f} / {float(e_pos[-edge:].sum() / e_pos.sum()):.3f}")
        # per scale
        for s in cfg.stft_scales:
            c1 = O.v2_config(stft_scales=(s,))
            yy = ys.detach().cpu().double().requires_grad_(True)
            O.audio_distance_v1(xs.detach().cpu().double(), yy, c1).backward()
            from rave_amd import ops as R
            yg = ys.detach().clone().requires_grad_(True)
            w = [torch.hann_window(s, device=dev)]
            R.multiscale_stft_distance(xs.reshape(-1, xs.shape[-1]), yg.reshape(-1, ys.shape[-1]), w, [s], 1e-7).backward()
            print(f"   scale {s:5d}: fused d/dy vs cpu_f64 rel-L2 {rel(yg.grad, yy.grad):.3e}")


if __name__ == "__main__":
    main()
