"""Why the shipped step leaves the reference step's trajectory after ONE optimizer step.

``profiles/round4_reference_training_step_on_mi355x.log``: the unmodified reference ``training_step`` on the drop-ins (A) and
``rave_amd.model`` as shipped (B2: spectral distance inside ``stft_loss.hip``) log the same step-0 losses to 7 digits, yet
after the first Adam step their parameters differed by up to 0.54 of the update.  Round 4 offered "Adam's first step turns a
sign change of a near-zero gradient element into +-lr" and left it untested.  This test measures it, and what it found is in
``profiles/round5_step0_gradient_signs.txt`` / ``profiles/round5_stft_pair_equalisation.txt``:

* the two steps differ ONLY in how the multi-scale spectral distance and its gradient are evaluated (torch.stft + autograd
  against the in-kernel transform), and Adam's first step is ``lr * g / (|g| + 1e-8)`` = ``lr * sign(g)``: an element whose
  sign differs moves by ``2 lr`` the other way however small it is, so ``|A - B| / |update| = 2 sqrt(fraction of flipped signs)``;
* **the round-4 kernel was not in the reference's noise class**: it transformed a frame of x and a frame of y as one complex
  transform, whose rounding error scales with the LOUDER of the two; at step 0 the decoder's output is ~ 50x below the target,
  and d log(|Y| + 1e-7) amplified the inherited noise into a multiband loss gradient 60 % away from the f64 value (torch's
  separate f32 transforms: 0.3 %).  9 % of all parameter-gradient elements had the other sign (14 % in the decoder), at
  |g| up to 2000 sigma, rms(S - R64) = 170 sigma: that is the 0.54;
* with the frame pair equalised by a power of two (``frame_scale``) the kernel is in the class of separate f32 transforms:
  rms(S - R64) = 3 - 4 sigma, 0.2 % of the elements change sign, all of them below the noise level.

The hot path runs on the HIP kernels once per cotangent set; the loss cotangents at its outputs come from (S) the shipped
fused loss, (R32) the reference's loss arithmetic in f32 (the oracle's ``audio_distance_v1`` = torch.stft + ATen, CPU),
(R64) the same in f64.  ``sigma = rms(g_R32 - g_R64)`` per tensor is the reference gradient's own f32 noise.  Asserted per tensor:

* the shipped gradient's distance from f64 is of the order of the reference's own: ``rms(S - R64) <= 6 sigma``;
* sign changes occur only below the noise level of the two f32 evaluations, ``nu = max(sigma, rms(S - R64))``: at most 2 % of
  the flipped elements of a tensor (or 3 elements) exceed ``4 nu`` and none ``16 nu``;
* overall less than 1 % of the elements change sign (the round-4 kernel: 8.9 %).

The table goes to ``gpurun_out/step0_gradient_signs.txt`` (committed copy: ``profiles/round5_step0_gradient_signs.txt``).
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rave_oracle as O  # noqa: E402  (the checker: tests only)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    return torch.device("cuda:0")


def test_step0_gradient_sign_changes_vs_the_reference_loss_lie_below_its_f32_noise(dev):
    from rave_amd import model as M
    batch = 4                                   # the geometry of the committed log: batch 4 x 65536, CAPACITY 96
    cfg = O.v2_config()
    sd = O.init_state_dict(cfg, seed=0, with_discriminator=False)
    x = O.synthetic_batch(batch, 1, 65536, seed=300)
    eps = torch.randn(batch, 128, 32, generator=torch.Generator().manual_seed(1000))
    m = M.build_v2()
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).train()
    xd, epsd = x.to(dev), eps.to(dev)

    def hot_path():
        m.zero_grad(set_to_none=True)
        m.prepare_weights()
        zp, x_mb = m.encode(xd, return_mb=True)
        z, reg = m.encoder.reparametrize(zp, epsd)[:2]
        y_mb = m.decoder(z)
        y_raw = m.decode(z)[..., :65536]
        return x_mb, y_mb[..., :4096], y_raw, reg

    # ---- loss cotangents at the hot-path outputs: shipped fused loss (S), reference arithmetic in f32 / f64 (R32 / R64)
    x_mb, y_mb, y_raw, reg = hot_path()
    ys, ym = y_raw.detach().clone().requires_grad_(True), y_mb.detach().clone().requires_grad_(True)
    loss_s = sum(m.audio_distance(xd, ys).values()) + sum(m.multiband_audio_distance(x_mb.detach(), ym).values())
    loss_s.backward()
    cots = {"S": (ys.grad.clone(), ym.grad.clone())}
    losses = {"S": float(loss_s)}
    for tag, dt in (("R32", torch.float32), ("R64", torch.float64)):
        yr, yb = y_raw.detach().cpu().to(dt).requires_grad_(True), y_mb.detach().cpu().to(dt).requires_grad_(True)
        loss_r = O.audio_distance_v1(x.to(dt), yr, cfg) + O.audio_distance_v1(x_mb.detach().cpu().to(dt), yb, cfg)
        loss_r.backward()
        cots[tag] = (yr.grad.float().to(dev), yb.grad.float().to(dev))
        losses[tag] = float(loss_r)
    assert abs(losses["S"] - losses["R64"]) <= 2e-5 * abs(losses["R64"]), losses      # the VALUES agree (as the log shows)
    m.release_weights()

    # ---- the same hot-path backward under each cotangent set (forward is bit-deterministic: re-run, nothing retained)
    grads = {}
    for tag, (c_raw, c_mb) in cots.items():
        x_mb, y_mb, y_raw, reg = hot_path()
        torch.autograd.backward([y_raw, y_mb, reg], [c_raw, c_mb, torch.ones((), device=dev)])
        torch.cuda.synchronize()
        grads[tag] = {k: p.grad.detach().double().cpu().reshape(-1) for k, p in m.named_parameters()
                      if p.grad is not None and k.startswith(("encoder.", "decoder."))}
        m.release_weights()
    assert len(grads["S"]) == 112

    rows, worst_sigma, worst_sep, bad = [], 0.0, 0.0, []
    tot_flip = tot = 0
    for k in sorted(grads["S"]):
        gs, g32, g64 = grads["S"][k], grads["R32"][k], grads["R64"][k]
        sigma = float((g32 - g64).pow(2).mean().sqrt())
        err_s = float((gs - g64).pow(2).mean().sqrt())
        flip = (torch.sign(gs) != torch.sign(g32))
        f = float(flip.double().mean())
        nu = max(sigma, err_s, 1e-300)
        size = float((g32[flip].abs().max() / nu)) if bool(flip.any()) else 0.0
        n_flip = int(flip.sum())
        above4 = int((g32[flip].abs() > 4.0 * nu).sum()) if n_flip else 0
        # f32-vs-f64 sign flips of the REFERENCE's own gradient, for scale
        f_ref = float((torch.sign(g32) != torch.sign(g64)).double().mean())
        rel_noise = sigma / float(g64.pow(2).mean().sqrt())
        sep = 2.0 * f ** 0.5
        rows.append((k, gs.numel(), f, f_ref, size, rel_noise, err_s / max(sigma, 1e-300), sep))
        tot_flip += int(flip.sum()); tot += gs.numel()
        worst_sigma = max(worst_sigma, size)
        worst_sep = max(worst_sep, sep)
        if size > 16.0 or above4 > max(3, 0.02 * n_flip) or err_s > 6.0 * sigma + 1e-12:
            bad.append((k, size, above4, err_s / max(sigma, 1e-300)))
    head = (f"step-0 parameter gradients, v2 CAPACITY 96, batch {batch} x 65536: shipped fused spectral loss (S) vs the reference's loss "
            f"arithmetic in f32 (R32) / f64 (R64), same HIP hot path\n"
            f"loss values: S {losses['S']:.7f}  R32 {losses['R32']:.7f}  R64 {losses['R64']:.7f}\n"
            f"sigma = rms(g_R32 - g_R64) per tensor = the reference gradient's own f32 noise; nu = max(sigma, rms(g_S - g_R64))\n"
            f"{'tensor':58s} {'numel':>9s} {'sign S!=R32':>11s} {'sign R32!=R64':>13s} {'max|g|/nu @flip':>17s} {'sigma/rms(g)':>12s} "
            f"{'rms(S-R64)/sigma':>16s} {'2sqrt(f)':>8s}")
    lines = [head] + [f"{k:58s} {n:9d} {f:11.4f} {fr:13.4f} {sz:17.2f} {rn:12.2e} {es:16.2f} {sp:8.3f}"
                      for k, n, f, fr, sz, rn, es, sp in rows]
    lines.append(f"all tensors: {tot_flip} of {tot} elements change sign ({tot_flip / tot:.4%}); largest flipped |g| = {worst_sigma:.2f} nu; "
                 f"largest predicted |A - B| / |update| after one Adam step (lr * sign(g)): {worst_sep:.2f} "
                 f"(the round-4 kernel, without the pair equaliser: 8.9 % of the elements, 0.80 predicted, 0.54 measured on the reference step, "
                 f"profiles/round4_reference_training_step_on_mi355x.log)")
    text = "\n".join(lines)
    print(text)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "step0_gradient_signs.txt"), "w") as fh:
        fh.write(text + "\n")
    assert not bad, bad[:5]
    assert 0 < tot_flip < 0.01 * tot, (tot_flip, tot)
