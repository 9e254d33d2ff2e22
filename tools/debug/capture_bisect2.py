"""Second bisect of the discrete GAN-phase capture crash (hipStreamEndCapture segfault): which STEP KIND and which
DISCRIMINATOR make the recording die?  One variant per process.
   python tools/debug/capture_bisect2.py            (driver)      python tools/debug/capture_bisect2.py worker <variant>"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = ["spectral_dis_step", "spectral_gen_step", "v2disc_dis_step", "v2disc_gen_step"]
# third level (the crash needs neither the RVQ -- off in these runs -- nor the spectral discriminator): the v2 config, which
# records, moved towards the discrete one a keyword at a time (generator step)
KW = {"kw_v2": {}, "kw_ratios": dict(ratios=(4, 4, 2, 2)), "kw_discrete_enc": dict(encoder_kind="discrete"),
      "kw_noise128": dict(noise_augmentation=128), "kw_skip0": dict(num_skipped_features=0), "kw_logeps1": dict(log_epsilon=1.0),
      "kw_enc_noise": dict(encoder_kind="discrete", noise_augmentation=128),
      "kw_all_but_noise": dict(ratios=(4, 4, 2, 2), encoder_kind="discrete", log_epsilon=1.0, num_skipped_features=0)}
VARIANTS += list(KW)
if len(sys.argv) > 2 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    import torch
    from rave_amd import model as M
    v = sys.argv[2]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    kw = {}
    if v.startswith("v2disc"):
        kw["spectral"] = False
    if v in KW:
        m = M.build_v2(**KW[v]).to(dev).train()
    else:
        m = M.build_discrete(**kw).to(dev).train()
    if "1scale" in v:         # keep only the first STFT scale of the spectral discriminator
        d = m.discriminator
        for holder in [d] + list(d.modules()):
            for name in ("nets", "discriminators", "layers"):
                lst = getattr(holder, name, None)
                if isinstance(lst, torch.nn.ModuleList) and len(lst) > 1 and "Spectral" in type(holder).__name__:
                    setattr(holder, name, torch.nn.ModuleList(list(lst)[:1]))
                    for a in ("scales", "n_ffts"):
                        if hasattr(holder, a):
                            setattr(holder, a, list(getattr(holder, a))[:1])
    m.configure_optimizers(capturable=True)
    m.warmed_up = True
    x = (0.3 * torch.randn(8, 1, 65536)).clamp(-1, 1).to(dev)
    import contextlib
    pre = os.environ.get("PRE", "default")      # where the eager pre-steps run: default stream / a side stream / not at all
    ctx = torch.cuda.stream(torch.cuda.Stream()) if pre == "side" else contextlib.nullcontext()
    if pre != "none":
        with ctx:
            for i in range(2):            # eager steps of both kinds: codebooks initialised
                m.training_step(x.clone(), i, capture_safe=(os.environ.get("PRE_SAFE", "1") == "1"))
                m.on_train_batch_end(None, None, i)
    torch.cuda.synchronize()
    idx = 0 if "dis_step" in v else 1
    if len(sys.argv) > 3:
        VARIANTS = sys.argv[3:]
    # which batch index is which kind?  (rave/model.py:288-296: the discriminator trains on even steps once warmed up)
    g = M.GraphedTrainingStep(m, x)
    print(v, "capturing batch_idx", idx, flush=True)
    g(x, idx)
    torch.cuda.synchronize()
    g(x, idx)
    torch.cuda.synchronize()
    print(v, "CAPTURED AND REPLAYED", {k: float(t) for k, t in list(g.logged.items())[:3]}, flush=True)
else:
    for v in (sys.argv[1:] or VARIANTS):
        r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), "worker", v], capture_output=True, text=True, timeout=400)
        tail = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.strip()][-3:]
        print(f"{v}: rc={r.returncode}  " + " | ".join(t[:160] for t in tail), flush=True)
