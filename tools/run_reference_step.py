"""The UNMODIFIED reference ``rave.model.RAVE.training_step`` (rave/model.py:288-424) executed on the HIP drop-ins ON THE
MI355X.  Builder-side measurement, not product code and not a test the driver runs: the reference
package cannot travel, so ``tools/run_reference_step.sh`` stages a read-only copy of /root/reference/rave under the
git-ignored ``oracle/_ref/reference`` for this one ``gpurun`` call and removes it afterwards.

  A  = the reference's own ``rave.RAVE`` class built from its own configs/v2.gin + rave_amd/configs/mi355x.gin (the
       overlay of INTEGRATION.md: pqmf / encoder / decoder / discriminator are the drop-in modules), its own
       ``training_step`` / ``configure_optimizers``, its own loss modules (rave.core.AudioDistanceV1 over torch.stft);
  B1 = ``rave_amd.model.RAVE`` (the restatement that runs where the reference cannot) with A's loss classes swapped in:
       the two training_step bodies then launch the same operators -> parameters expected bit-equal (up to whatever
       run-to-run nondeterminism torch's own stft backward has: A is also run twice to measure that);
  B2 = ``rave_amd.model.RAVE`` as shipped (spectral distance / feature matching on the fused HIP kernels): same
       trajectory up to the conditioning of the loss gradient.

VAE phase and GAN phase (discriminator step + generator steps), ``STEPS`` steps each from the same seeded weights, same
seeded noise.  Prints logged losses per step and parameter differences relative to the size of the update.
"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.dont_write_bytecode = True
os.environ.setdefault("RAVE_REFERENCE_ROOT", os.path.join(ROOT, "oracle", "_ref", "reference"))

import torch  # noqa: E402

from ref_import import import_reference  # noqa: E402
from ref_models import attach_optimizers  # noqa: E402

STEPS = int(os.environ.get("STEPS", "4"))
# torch's own stft / abs / fold backward kernels accumulate with atomics: two runs of the SAME reference step differ in the
# last bits, and the chaotic first steps of training (batch 4, random init, Adam at 1e-3) amplify that to O(1) within 3
# steps (run 1 of this script: profiles/round4_reference_training_step_on_mi355x_run1.log).  DETERMINISTIC=1 (default)
# asks torch for its deterministic kernels so that "A run twice" is bit-equal and A vs B1 can be read as an identity check.
if os.environ.get("DETERMINISTIC", "1") == "1":
    os.environ.setdefault("CUBLAS_WORKSPACE_CONFIG", ":4096:8")
BATCH = int(os.environ.get("BATCH", "4"))
CAP = os.environ.get("CAPACITY")          # default: the real v2 width (96)


def main():
    if os.environ.get("DETERMINISTIC", "1") == "1":
        torch.use_deterministic_algorithms(True, warn_only=True)
    dry = "--dry" in sys.argv            # build container (no GPU): construct A / B1 / B2 and check the state_dict hand-over
    dev = torch.device("cpu" if dry else "cuda:0")
    rave = import_reference()
    import gin
    import rave_oracle as O
    from rave_amd import model as M

    def build_a():
        gin.clear_config()
        binds = [f"CAPACITY = {CAP}"] if CAP else []
        gin.parse_config_files_and_bindings(["configs/v2.gin", os.path.join(ROOT, "rave_amd", "configs", "mi355x.gin")], binds)
        torch.manual_seed(0)
        m = rave.RAVE()
        assert type(m).__module__ == "rave.model"
        mods = {k: type(getattr(m, k)).__module__ for k in ("pqmf", "encoder", "decoder", "discriminator")}
        assert all(v.startswith("rave_amd.") for v in mods.values()), mods
        convs = {type(x).__module__ for x in m.modules() if "Conv" in type(x).__name__}
        assert all(c.startswith("rave_amd") for c in convs), convs      # no torch / cached_conv convolution in the model
        return m.to(dev).train()

    a0 = build_a()
    print("A:", type(a0).__module__ + "." + type(a0).__name__, "training_step from", type(a0).training_step.__module__,
          "| audio_distance", type(a0.audio_distance).__module__, "| encoder", type(a0.encoder).__module__)
    sd0 = {k: v.detach().clone() for k, v in a0.state_dict().items()}
    cap = int(CAP) if CAP else 96

    def build_b(ref_losses):
        torch.manual_seed(0)
        b = M.build_v2(capacity=cap)
        missing = b.load_state_dict({k: v for k, v in sd0.items()}, strict=False)
        hot = [k for k in missing.missing_keys if k.startswith(("pqmf.", "encoder.", "decoder.", "discriminator."))]
        assert not hot, hot[:5]
        if ref_losses:
            b.audio_distance = copy.deepcopy(a0.audio_distance)
            b.multiband_audio_distance = copy.deepcopy(a0.multiband_audio_distance)
            b.feature_matching_fun = a0.feature_matching_fun
            b.gan_loss = a0.gan_loss
        b = b.to(dev).train()
        if ref_losses:
            # the reference's optimizers, object for object (rave/model.py:226-236): B1 isolates the training_step body
            gen_p = list(b.encoder.parameters()) + list(b.decoder.parameters())
            gen_opt = torch.optim.Adam(gen_p, 1e-3, (.5, .9))
            dis_opt = torch.optim.Adam(list(b.discriminator.parameters()), 1e-4, (.5, .9))
            b._opts = (gen_opt, dis_opt)
            b._gen_sched = torch.optim.lr_scheduler.LinearLR(gen_opt, start_factor=1.0, end_factor=0.1, total_iters=b.warmup)
        else:
            b.configure_optimizers()
        return b

    if dry:
        build_b(True), build_b(False), attach_optimizers(build_a())
        print("dry run OK:", len(sd0), "state_dict entries handed from rave.RAVE to rave_amd.model.RAVE")
        return
    xs = [O.synthetic_batch(BATCH, 1, 65536, seed=300 + i).to(dev) for i in range(STEPS)]

    def run(model, is_ref, warmed_up, fm_fused=True):
        """Returns (parameters after EVERY step, logged losses per step)."""
        model.warmed_up = warmed_up
        logs, snaps = [], []
        os.environ["RH_FM_FUSED"] = "1" if fm_fused else "0"
        for i in range(STEPS):
            torch.manual_seed(1000 + i)
            model.training_step(xs[i].clone(), i)
            model.on_train_batch_end(None, None, i)
            logs.append({k: float(v) for k, v in model.logged.items() if torch.is_tensor(v) or isinstance(v, float)})
            torch.cuda.synchronize()
            snaps.append({k: v.detach().clone() for k, v in model.named_parameters()})
        return snaps, logs

    def fresh_a():
        m = build_a()
        m.load_state_dict(sd0)
        attach_optimizers(m)
        return m

    def diff(sa, sb, p0, label):
        parts = []
        for i, (pa, pb) in enumerate(zip(sa, sb)):
            worst_upd, nbit = 0.0, 0
            for k in pa:
                d = (pa[k].double() - pb[k].double()).norm()
                if float(d) == 0.0:
                    nbit += 1
                    continue
                upd = (pa[k].double() - p0[k].double()).norm().clamp_min(1e-30)
                worst_upd = max(worst_upd, float(d / upd))
            parts.append(f"step {i}: {nbit}/{len(pa)} bit-equal, worst |A-B|/|update| {worst_upd:.1e}")
        print(f"   {label}: " + "; ".join(parts))

    p0 = {k: v.detach().clone() for k, v in fresh_a().named_parameters()}
    for phase, warmed in (("VAE phase", False), ("GAN phase", True)):
        print(f"== {phase}: {STEPS} steps, batch {BATCH} x 65536, capacity {cap}")
        pa, la = run(fresh_a(), True, warmed)
        pa2, la2 = run(fresh_a(), True, warmed)
        pb1, lb1 = run(build_b(True), False, warmed)
        pb2, lb2 = run(build_b(False), False, warmed)
        keys = sorted(set(la[0]) & set(lb1[0]))
        for i in range(STEPS):
            print(f"   step {i}: " + "  ".join(f"{k} A {la[i][k]:.7g} | B1 {lb1[i].get(k, float('nan')):.7g} | B2 {lb2[i].get(k, float('nan')):.7g}"
                                             for k in keys if k != "beta_factor"))
        same_logs = all(la[i].get(k) == lb1[i].get(k) for i in range(STEPS) for k in keys)
        print(f"   logged losses A == B1 at every step (exact float equality): {same_logs}")
        moved = sum(1 for k in pa[-1] if not torch.equal(pa[-1][k], p0[k]))
        print(f"   parameters moved by A: {moved} of {len(pa[-1])}")
        diff(pa, pa2, p0, "A run twice (torch's own run-to-run nondeterminism)")
        diff(pa, pb1, p0, "A vs B1 (rave_amd.model.training_step, reference loss modules)")
        diff(pa, pb2, p0, "A vs B2 (rave_amd.model as shipped: fused HIP losses)")


if __name__ == "__main__":
    main()
