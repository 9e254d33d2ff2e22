#!/usr/bin/env python
"""bench.py -- RAVE v2 training step on MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--phase vae|gan]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete ``training_step`` (rave/model.py:288-413) of the v2 config on a synthetic
minibatch already resident in HBM: PQMF analysis -> EncoderV2 -> reparametrize -> GeneratorV2 ->
PQMF synthesis -> multi-scale STFT losses -> backward -> Adam.  Default workload = BASELINE.json
configs[1]: v2, batch 32 mono, 44.1 kHz, n_signal 65536, VAE phase.  Multi-GPU: weak scaling, the
minibatch is sharded (32 clips per GPU), gradients averaged with bucketed RCCL all-reduce
overlapped with backward.  Rank 0 prints ONE JSON line.  ``--gpus N`` without a torchrun environment
re-launches itself under ``torch.distributed.run`` with N ranks (and fails if fewer GPUs are visible).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# SURVEY.md section 8d contract figures for v2 forward (PQMF -> enc -> dec -> PQMF^-1), per mono clip
V2_FWD_MAC_PER_CLIP = 4_786_814_976
V2_FWD_ACT_BYTES_PER_CLIP = 83_673_088
V2_WEIGHT_BYTES = 126_123_072
HBM_PEAK = 8.0e12          # B/s   MI355X_MICROARCH.md chip-level parameters
F32_MFMA_PEAK = 157.3e12   # FLOP/s (f32-input MFMA == f32 vector peak)
# FLOP/s f32-equivalent of the 16-bit matrix cores (2.5 PF dense) for the x6 kernels.  Round 6 (the product build, RH_X6_F16 = 1):
# two f16 pieces per operand, THREE v_mfma_f32_32x32x16_f16 per product block -> 2.5 PF / 3 = 833 TFLOP/s; the comparison
# build (rounds 2-5): three bf16 pieces, SIX products -> 417.  main() picks by what the loaded library says.
X6_PRODUCTS = 3
X6_MFMA_PEAK = 2.5e15 / 3
# what a loop of nothing but these MFMAs sustains on this part with random operands (two waves per SIMD): the chip is
# POWER-limited there, the shader clock reads ~1.6 GHz instead of 2.4 -- bf16: 17.9 ns per MFMA and SIMD
# (tools/probe/mfma_clock.hip, profiles/round4_probe_mfma_clock.txt), f16: 19.5 ns (tools/probe/f16x3.hip,
# profiles/round6_probe_f16x3.txt: the f16 multipliers draw ~9 % more)
X6_MFMA_SUSTAINED = 32768 * 1024 / 19.5e-9 / 3


def _set_x6_mode(f16: bool) -> None:
    global X6_PRODUCTS, X6_MFMA_PEAK, X6_MFMA_SUSTAINED
    X6_PRODUCTS = 3 if f16 else 6
    X6_MFMA_PEAK = 2.5e15 / X6_PRODUCTS
    X6_MFMA_SUSTAINED = 32768 * 1024 / (19.5e-9 if f16 else 17.9e-9) / X6_PRODUCTS


def _cpu_point(batch: int, threads: int, n_signal: int) -> None:
    """Child process of cpu_baseline (``bench.py --cpu-point B,THREADS,N``): the oracle's v2 VAE-phase training step (forward +
    losses + backward + Adam) at one (batch, threads) point on the host cores; prints ``CPU_POINT {json}``.  Never touches the
    GPU and imports nothing of the product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rave_oracle as O
    torch.set_num_threads(threads)
    cfg = O.v2_config()
    sd = O.init_state_dict(cfg, seed=0, with_discriminator=False)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(("encoder.", "decoder."))}
    full = dict(sd)
    full.update(leaves)
    opt = torch.optim.Adam(list(leaves.values()), 1e-3, (.5, .9))
    x = O.synthetic_batch(batch, 1, n_signal)
    eps = torch.randn(batch, cfg.latent_size, n_signal // 2048)

    def one():
        opt.zero_grad()
        xx = x.clone().requires_grad_(True)
        loss, _, _, _ = O.generator_losses(xx, full, cfg, eps, warmed_up=False)
        loss.backward()
        opt.step()

    t = time.perf_counter()
    one()                                            # warm-up (oneDNN primitive creation)
    warm = time.perf_counter() - t
    print("CPU_POINT " + json.dumps({"warm_s": warm}), flush=True)      # (a parent that has to kill us still learns this much)
    ts = []
    for _ in range(2 if warm > 1.0 else 3):
        t = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t)
        med = sorted(ts)[len(ts) // 2]              # (reported after every step: a point killed at its limit keeps what it measured)
        print("CPU_POINT " + json.dumps({"warm_s": warm, "step_s": med, "steps_timed": len(ts), "threads_used": torch.get_num_threads()}),
              flush=True)
    os._exit(0)                                     # (skip the interpreter's teardown of the thread pools: seconds on a big host)


def _host_cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            models = [ln.split(":", 1)[1].strip() for ln in f if ln.lower().startswith("model name")]
        return f"{models[0]} ({len(models)} hardware threads)" if models else "unknown"
    except OSError:
        return "unknown"


def cpu_baseline(n_signal: int, budget_s: float = 48.0):
    """The oracle (CPU fp32 ATen restatement of the reference, oracle/rave_oracle.py -- bit-pinned to the reference
    modules, tests/test_oracle.py) timed on this box's host cores on a bounded sample of the same workload (SURVEY section 8d:
    ``torch.set_num_threads`` up to nproc, the host's model string recorded):
    (1) the metric's unit of work, the v2 VAE-phase training step (fwd + losses + bwd + Adam): a THREAD SWEEP 16 / 32 / 64 /
        128 / ... / nproc at batch 8, then the configs[1] batch (32) at the best thread count; best point reported;
    (2) BASELINE configs[0] as worded: v2_small, 1 mono clip, forward + loss only.
    Every point runs in its own child process under a timeout: with every hardware thread of a 256-thread host oneDNN + OpenMP
    ran this model > 100x slower than with 16-32 (round 2: one batch-8 step did not finish in 5 minutes) -- such a point is
    killed and recorded as timed out instead of taking the bench with it."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rave_oracle as O
    nproc = os.cpu_count() or 1
    t_start = time.perf_counter()
    left = lambda: budget_s - (time.perf_counter() - t_start)

    def timed(fn, n):
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        ts.sort()
        return ts[len(ts) // 2]

    def point(b, thr, limit):
        rec = {"batch": b, "threads": thr}
        if limit < 3.0:
            rec["skipped"] = "budget spent"
            return rec
        env = dict(os.environ)
        env.pop("OMP_NUM_THREADS", None)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-point", f"{b},{thr},{n_signal}"],
                               capture_output=True, text=True, timeout=limit, env=env)
            out = r.stdout
        except subprocess.TimeoutExpired as e:
            out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            rec["timed_out_after_s"] = round(limit, 1)
        last = [ln for ln in out.splitlines() if ln.startswith("CPU_POINT ")]
        if last:
            rec.update(json.loads(last[-1][len("CPU_POINT "):]))
        if "step_s" in rec:
            rec["samples_per_s"] = b * n_signal / rec["step_s"]
        return rec

    sweep = sorted({t for t in (16, 32, 64, 128, 256, nproc) if t <= nproc} | {min(nproc, 16)})
    points = []
    # batch 8: ~0.3-1 s per step at the rates measured so far.  The small thread counts first, then BASELINE configs[1]'s batch
    # at the best of them, then the large counts with what is left of the budget (on the 256-thread EPYC hosts of this pool 64
    # threads already run 4x slower than 16, and 128 / 256 do not finish a step in 9 s: they are killed at their limit)
    low = [t for t in sweep if t <= 64]
    for thr in low:
        points.append(point(8, thr, min(8.0, left() - 20.0)))
    done = [p for p in points if "samples_per_s" in p]
    if done:
        best_thr = max(done, key=lambda p: p["samples_per_s"])["threads"]
        points.append(point(32, best_thr, min(12.0, left() - 6.0)))          # BASELINE configs[1]'s batch
    for thr in [t for t in sweep if t > 64]:
        points.append(point(8, thr, min(6.0, left() - 3.0)))
    done = [p for p in points if "samples_per_s" in p]
    best = max(done, key=lambda p: p["samples_per_s"]) if done else None

    # (2) configs[0]: v2_small forward + loss, one clip
    small = None
    base_thr = min(nproc, 16)
    if left() > 2.0:
        try:
            cs = O.v2_small_config()
            sds = O.init_state_dict(cs, seed=0, with_discriminator=False)
            xs = O.synthetic_batch(1, 1, n_signal)
            hop = cs.n_band
            for r in cs.ratios:
                hop *= r
            eps_s = torch.randn(1, cs.latent_size, n_signal // hop)
            torch.set_num_threads(base_thr)

            def small_fwd():
                with torch.no_grad():
                    O.generator_losses(xs, sds, cs, eps_s, warmed_up=False)

            small_fwd()
            ts = timed(small_fwd, 5)
            small = {"value": n_signal / ts, "unit": "samples/s", "ms": 1e3 * ts, "cores": base_thr,
                     "sample": f"BASELINE configs[0]: v2_small, 1 mono clip x {n_signal}, forward + loss, no_grad, median of 5"}
        except Exception as e:   # the headline never depends on the auxiliary baseline
            small = {"error": repr(e)}
    res = {"value": best["samples_per_s"] if best else None, "unit": "samples/s", "cores": best["threads"] if best else None,
           "host_cores": nproc, "host_cpu": _host_cpu_model(), "kind": "port",
           "ms_per_step": 1e3 * best["step_s"] if best else None,
           "sample": (f"v2 VAE-phase training step (fwd+losses+bwd+Adam) x {n_signal} samples per clip, median step time after one "
                      f"warm-up step; thread sweep {sweep} at batch 8 + batch 32 (BASELINE configs[1]) at the best thread count, one "
                      f"child process per point under a timeout; best = batch {best['batch'] if best else None}, "
                      f"{best['threads'] if best else None} threads; torch CPU fp32 oracle (oracle/rave_oracle.py); bounded to "
                      f"~{budget_s:.0f} s"),
           "points": points, "v2_small_forward_loss": small}
    return res


def _lib_sha():
    import hashlib
    from rave_amd import _lib as L
    with open(L.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def _pmc_traffic(kernel_name):
    """(HBM bytes per launch of the dominant kernel family, note) from the newest committed PMC summary.  SELF-CHECKING
    (VERDICT r3 #8): tools/final_measure.sh stamps the summary with the sha256 of the librave_hip.so the counters were
    collected on; a different library is loaded now -> traffic null + a note (the figure would silently be stale)."""
    fam = {"conv_x6_kernel": "conv_x6(fwd+dgrad)", "wgrad_x6_kernel": "wgrad_x6", "wgrad_dma_kernel": "wgrad_f32",
           "conv_igemm_dma_kernel": "conv_f32(fwd+dgrad)"}.get(kernel_name)
    for name in ("round6_pmc_traffic.json", "round5_pmc_traffic.json", "round4_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                d = json.load(f)
            sha = d.get("librave_hip_sha256")
            if sha is None:
                return None, f"profiles/{name} carries no library stamp (collected before round 4): not used"
            if sha != _lib_sha():
                return None, (f"profiles/{name} was collected on librave_hip.so sha256 {sha[:16]}..., the loaded library is "
                              f"{_lib_sha()[:16]}...: stale, not reported -- re-run tools/final_measure.sh (PARTS=pmc)")
            return float(d[fam]["hbm_bytes_per_launch"]), f"profiles/{name}, library stamp verified ({sha[:16]}...)"
        except Exception as e:                      # noqa: BLE001
            return None, f"profiles/{name}: {e!r}"
    return None, "no PMC summary committed"


def _forward_leg(batch: int, n_signal: int) -> None:
    """Child process (``bench.py --forward-leg B,N`` with RAVE_HIP_LIB pointing at a library): the inference forward of the
    north-star leg on that library; prints ``FORWARD_LEG {json}``."""
    from rave_amd import _lib as L, model as M
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = M.build_v2().to(dev).train()
    x = 0.3 * torch.randn(batch, 1, n_signal, generator=torch.Generator().manual_seed(1)).clamp(-1, 1).to(dev)
    with torch.no_grad():
        m.prepare_weights(reuse=True)

        def once():
            m.prepare_weights(reuse=True)
            m.decode(m.encoder.reparametrize(m.encode(x))[0])
            m.release_weights()
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            once()
        e1.record()
        torch.cuda.synchronize()
    print("FORWARD_LEG " + json.dumps({"ms": e0.elapsed_time(e1) / 10, "library": os.path.basename(L.LIB_PATH),
                                       "f16_pieces": int(L.lib.rh_x6_uses_ranges())}))


def bf16x6_leg(batch: int, n_signal: int):
    """SECONDARY key forward_only_bf16x6: the forward leg on the comparison build of the x6 kernels in their round 2-5 form
    (three bf16 pieces per operand, six partial products: rave_amd/_var/librave_hip_bf16.so, RH_X6_F16 = 0), in its own
    process -- what the round-6 change to two f16 pieces / three products bought, on the same box."""
    import subprocess
    from rave_amd import build as B
    lib = B.variant_lib()
    if not os.path.exists(lib):
        return {"forward_only_bf16x6": {"error": f"{lib} missing: python -m rave_amd.build --variant"}}
    try:
        env = dict(os.environ, RAVE_HIP_LIB=lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--forward-leg", f"{batch},{n_signal}"], capture_output=True,
                           text=True, timeout=300, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("FORWARD_LEG ")]
        if r.returncode != 0 or not line:
            return {"forward_only_bf16x6": {"error": (r.stderr or r.stdout)[-400:]}}
        d = json.loads(line[-1][len("FORWARD_LEG "):])
        d["dtype"] = "f32 as 3 bf16 pieces x 6 products (rounds 2-5)"
        return {"forward_only_bf16x6": d}
    except Exception as e:                          # noqa: BLE001 -- secondary keys never break the headline
        return {"forward_only_bf16x6": {"error": repr(e)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--n-signal", type=int, default=65536)
    ap.add_argument("--phase", choices=["vae", "gan"], default="vae")
    ap.add_argument("--config", choices=["v2", "discrete", "v3"], default="v2",
                    help="v2 = BASELINE configs[1]/[2] (the metric's config); discrete = configs[3] (RVQ + spectral "
                         "discriminator); v3 = configs[4] (stereo, causal, snake, descript discriminator)")
    ap.add_argument("--skip-dead-grads", action="store_true",
                    help="opt-in (NOT the headline): skip gradients the reference computes and discards "
                         "(rave_amd.model.RAVE.skip_dead_grads)")
    ap.add_argument("--no-graph", action="store_true",
                    help="run the eager step instead of replaying the captured hipGraph (data-parallel runs record the "
                         "bucket all-reduces into the same graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-graph", action="store_true",
                    help="(no-op since round 6: every config records its step; kept for old command lines)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-products-leg", action="store_true",
                    help="skip the secondary forward_only_bf16x6 leg (comparison build: 3 bf16 pieces, 6 products)")
    ap.add_argument("--forward-leg", default=None, help=argparse.SUPPRESS)   # child process of bf16x6_leg()
    ap.add_argument("--cpu-point", default=None, help=argparse.SUPPRESS)      # child process of cpu_baseline()
    args = ap.parse_args()
    if args.cpu_point:
        b, thr, n = (int(v) for v in args.cpu_point.split(","))
        return _cpu_point(b, thr, n)
    if args.forward_leg:
        b, n = (int(v) for v in args.forward_leg.split(","))
        return _forward_leg(b, n)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: one process per GPU, launched here (never a silent single-rank run)
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} requested but only {torch.cuda.device_count()} GPU(s) visible")
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus})")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("RAVE_FORCE_DIST", "0") == "1"   # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from rave_amd import _lib as _L, ddp, model as M, ops
    _set_x6_mode(_L.lib.rh_x6_uses_ranges() == 1)

    torch.manual_seed(0)
    n_ch = 1
    if args.config == "v2":
        m = M.build_v2()
    elif args.config == "discrete":
        m = M.build_discrete()
    else:
        m = M.build_v3()
        n_ch = 2
    m = m.to(dev).train()
    if args.config == "discrete":
        # QuantizeCallback never fires in the shipped trainer: enable the RVQ
        # explicitly so that rave/quantization.py is exercised; the k-means init runs in the warm-up steps
        m.encoder.enabled.fill_(1)
    ddp.broadcast_module(m)
    use_ddp = world > 1 or force_dist
    # (the discrete config initialises its RVQ codebooks with k-means inside its first training step: host-driven,
    # data-dependent work that a recorded graph cannot contain -- those steps run eagerly BEFORE the capture, below, and ON A
    # SIDE STREAM: rounds 4-5 ran them on the default stream and hipStreamEndCapture then segfaulted inside the runtime although
    # every component of the step captured on its own (profiles/round4_discrete_capture_bisect.txt).  Round 6
    # (tools/debug/capture_bisect2.py, profiles/round6_discrete_capture_bisect.txt): ANY eager GAN-phase step on the legacy
    # default stream before the recording -- the plain v2 config too -- kills the capture; the same steps on a side stream,
    # or none, and every config records.  --no-graph gives the eager step.)
    use_graph = not args.no_graph
    gen_opt, dis_opt = m.configure_optimizers(capturable=use_graph)
    m.warmed_up = args.phase == "gan"
    m.skip_dead_grads = bool(args.skip_dead_grads)
    gen_params = list(m.encoder.parameters()) + list(m.decoder.parameters())
    red_gen = ddp.GradReducer(gen_params, force=force_dist) if use_ddp else None
    red_dis = ddp.GradReducer(list(m.discriminator.parameters()), force=force_dist) if use_ddp and m.warmed_up else None
    # rank-0 buffers before every forward (torch DDP broadcast_buffers semantics): RVQ codebooks, BatchNorm stats
    bufsync = ddp.BufferSync(m, force=force_dist) if use_ddp else None

    # synthetic 44.1 kHz waveforms, one shard per rank, resident in HBM
    g = torch.Generator().manual_seed(20250509 + rank)
    t = torch.arange(args.n_signal, dtype=torch.float32) / 44100.0
    x = 0.1 * torch.randn(args.batch, n_ch, args.n_signal, generator=g)
    for f0, a in ((220.0, 0.2), (1760.0, 0.1), (7040.0, 0.05)):
        ph = torch.rand(args.batch, n_ch, 1, generator=g) * 6.283185307
        x = x + a * torch.sin(6.283185307 * f0 * t + ph)
    x = x.clamp(-1, 1).to(dev)

    # data-parallel hooks of one step: the reducer of the optimizer that steps (index 0 generator, 1 discriminator)
    ddp_exposed = []                      # (event before finish(), event after): communication the compute stream waits for

    def grad_begin(idx):
        (red_dis if idx == 1 else red_gen).begin()

    def grad_sync(idx):
        red = red_dis if idx == 1 else red_gen
        timed = ddp_exposed is not None and not torch.cuda.is_current_stream_capturing() and len(ddp_exposed) < 64
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        red.finish()
        if timed:
            e1.record()
            ddp_exposed.append((e0, e1))

    sync_kw = dict(grad_begin=grad_begin, grad_sync=grad_sync) if use_ddp else {}
    graphed = None
    graph_note = None
    if use_graph and args.config == "discrete":
        # eager steps until every codebook is initialised (one of each step kind): GraphedTrainingStep refuses an
        # un-initialised RVQ (rave_amd/model.py: _check_capturable)
        pre = torch.cuda.Stream()
        pre.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pre):          # NOT the default stream: see the note at use_graph
            for i in range(2 if m.warmed_up else 1):
                if use_ddp:
                    bufsync.sync()
                m.training_step(x.detach().clone(), i, capture_safe=True, **sync_kw)
                m.on_train_batch_end(None, None, i)
        torch.cuda.current_stream().wait_stream(pre)
        torch.cuda.synchronize()
    if use_graph:
        try:
            graphed = M.GraphedTrainingStep(m, x, before_step=(bufsync.sync if use_ddp else None), **sync_kw)
            graphed(x, 0)                 # capture now: a failure falls back to the eager step (reported)
            torch.cuda.synchronize()
        except Exception as e:            # noqa: BLE001 -- e.g. a collective the backend cannot record
            if not use_ddp and args.config == "v2":
                raise
            import traceback
            where = [ln.strip() for ln in traceback.format_exc().splitlines() if "rave_amd" in ln][-3:]
            graph_note = f"hipGraph capture of the step failed ({type(e).__name__}: {str(e)[:200]}); eager step; at {where}"
            graphed, use_graph = None, False
            m.configure_optimizers(capturable=False)

    def step(i, eager=False):
        if graphed is not None and not eager:
            graphed(x, i)
        elif use_ddp:
            bufsync.sync()
            m.training_step(x.detach().clone(), i, capture_safe=use_graph, **sync_kw)
        else:
            m.training_step(x.detach().clone(), i, capture_safe=use_graph)
        m.on_train_batch_end(None, None, i)       # generator LR schedule (rave/model.py:272-274)

    def fence():
        torch.cuda.synchronize()
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step(args.warmup + i)
        evs[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_ddp:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    samples = world * args.batch * args.n_signal * args.steps

    out = {
        "metric": f"audio samples/sec/GPU (RAVE {args.config} training step, 44.1 kHz, n_signal=65536)",
        "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config} config, batch {args.batch} {'stereo' if n_ch == 2 else 'mono'} 44.1 kHz "
                               f"n_signal={args.n_signal}, "
                               f"{'VAE' if args.phase == 'vae' else 'VAE+GAN'}-phase training step per GPU (--phase {args.phase})"
                               + (" [opt-in: dead gradients skipped]" if args.skip_dead_grads else ""),
                   "global_batch": world * args.batch, "parallelism": f"dp{world}", "phase": args.phase},
        "per_gpu_samples_per_s": samples / dt / world,
        "ms_per_step_median_hip_events": per_step[len(per_step) // 2] if per_step else None,
        "step_mode": "hipGraph replay (rave_amd.model.GraphedTrainingStep)" if use_graph else "eager",
    }
    if graph_note:
        out["step_mode_note"] = graph_note

    rec = None
    if not args.no_kernel_timing and args.config == "v2":
        # ---- per-launch HIP-event timing of the conv kernels over instrumented replays of the step.  EVERY rank runs
        # these steps (they contain the gradient all-reduce); rank 0 reports.
        reps = 2
        # (kernels are timed in ISOLATION: the weight-gradient side stream -- which overlaps launches in the timed
        # region above -- is switched off for these instrumented steps, so that an event pair brackets one kernel alone)
        side_env = os.environ.get("RH_BWD_SIDE_STREAM")
        os.environ["RH_BWD_SIDE_STREAM"] = "0"
        ops.profile_begin()
        for i in range(reps):
            step(args.warmup + args.steps + i, eager=True)     # per-launch events need the eager step
        rec = ops.profile_end()
        if side_env is None:
            os.environ.pop("RH_BWD_SIDE_STREAM", None)
        else:
            os.environ["RH_BWD_SIDE_STREAM"] = side_env
        fence()
    if rank == 0 and rec is not None:
        # per kind: [launches, flop, bytes, kernel ms (the main kernel's own start/stop timestamps), call ms (event bracket
        # around the whole C-ABI call: + split-K finalize / partial-sum reduction launches + event overhead)]
        agg = {}
        for kind, fl, by, kms, cms in rec:
            a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += fl; a[2] += by; a[3] += (kms if kms is not None else cms); a[4] += cms
        # every launch is priced against the peak of the instruction it issues: conv_x6_kernel = f32 as X6_PRODUCTS 16-bit
        # MFMAs per product block (round 6: three v_mfma_f32_32x32x16_f16 -> 2.5 PF / 3 = 833 TFLOP/s f32-equivalent;
        # comparison build: six bf16 ones -> 417); the f32-input MFMA
        # kernels (conv_igemm_dma_kernel forward / data gradient of the few geometries x6 does not take, and
        # wgrad_dma_kernel for the <= 96-row weight gradients on long sequences) = 157.3 TFLOP/s
        peak_of = lambda k: X6_MFMA_PEAK if k.endswith("[x6]") else F32_MFMA_PEAK
        kern_of = {"conv_fwd[x6]": "conv_x6_kernel", "conv_dgrad[x6]": "conv_x6_kernel",
                   "conv_fwd[f32]": "conv_igemm_dma_kernel", "conv_dgrad[f32]": "conv_igemm_dma_kernel",
                   "conv_wgrad[f32]": "wgrad_dma_kernel", "conv_wgrad[x6]": "wgrad_x6_kernel"}
        byk = {}
        for k, v in agg.items():
            a = byk.setdefault(kern_of.get(k, k), [0, 0.0, 0.0, 0.0, 0.0, peak_of(k)])
            for i in range(5):
                a[i] += v[i]
        dom_name = max(byk, key=lambda k: byk[k][3])
        n, fl, by, ms, cms, peak = byk[dom_name]
        out["roofline"] = {
            "bound": "mfma", "kernel": dom_name + (" (weight-gradient launches)" if dom_name.startswith("wgrad")
                                                   else " (forward + data-gradient launches; 6 of them per step are the fused "
                                                        "Residual(DilatedUnit) kernel unit_x6_kernel, same main loop)"),
            "achieved": fl / (ms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
            "frac": fl / (ms * 1e-3) / peak, "frac_vs_exact_f32_peak": fl / (ms * 1e-3) / F32_MFMA_PEAK,
            "frac_vs_sustained_mfma_rate": (fl / (ms * 1e-3) / X6_MFMA_SUSTAINED) if peak == X6_MFMA_PEAK else None,
            "traffic": _pmc_traffic(dom_name)[0],
            "traffic_source": _pmc_traffic(dom_name)[1],
            "traffic_note": "HBM bytes per launch (read + write) of this kernel family from rocprofv3 --pmc FETCH_SIZE / "
                            "WRITE_SIZE passes over this command (separate passes, tools/final_measure.sh), with read / write "
                            "factors calibrated on tools/probe/fetch_calib in the same call; committed under profiles/ with the "
                            "sha256 of the library it was collected on and read from there (the counters cannot be collected "
                            "inside the timed process); null when the loaded library differs from that stamp",
            "algorithmic_bytes_per_launch": by / n,
            "launches_per_step": n // reps, "avg_launch_ms": ms / n,
            "avg_call_ms": cms / n,
            "algorithmic_gflop_per_launch": fl / n / 1e9,
            "timing_note": "avg_launch_ms = the kernel's own duration: every launch is dispatched with a pair of HIP events as "
                           "its start / stop events (rh_set_kernel_events -> hipExtLaunchKernelGGL, on the launch stream), the "
                           "timestamps rocprofv3 reads -- compare profiles/round4_kernel_stats_step_b32.md (rocprofv3 "
                           "--kernel-trace --stats of `bench.py --no-graph`, RH_BWD_SIDE_STREAM=0).  Eager replays of the step "
                           "with the side stream off (one kernel at a time); the timed region itself overlaps the "
                           "weight-gradient branch on a second stream.  avg_call_ms = event bracket around the whole C-ABI "
                           "call: + the split-K finalize launch where the plan splits K, + event overhead",
            "note": "f32 in / f32 accumulate everywhere; peak = that of the instruction the kernel issues (conv_x6_kernel: "
                    + ("every f32 operand scaled by its tensor's power-of-two range and split into 2 f16 pieces, 3 "
                       "v_mfma_f32_32x32x16_f16 per product block -> 2.5 PF / 3 = 833 TFLOP/s f32-equivalent"
                       if X6_PRODUCTS == 3 else
                       "every f32 split exactly into 3 bf16, 6 v_mfma_f32_32x32x16_bf16 per product block -> 2.5 PF / 6 = "
                       "417 TFLOP/s f32-equivalent") + "; f32-input MFMA kernels: 157.3)",
            "x6_products": X6_PRODUCTS,
            "frac_vs_6_product_peak": fl / (ms * 1e-3) / (2.5e15 / 6),
            "hbm_frac": by / (ms * 1e-3) / HBM_PEAK,
            "bound_note": "arithmetic intensity of the family (algorithmic FLOP / algorithmic byte) vs the ridge of the issued "
                          "instruction (peak FLOP/s / 8 TB/s): %.0f vs %.0f FLOP/B" % (fl / by, peak / HBM_PEAK),
        }
        out["kernels"] = {k: ({"launches_per_step": v[0] // reps, "ms_per_step": v[3] / reps, "call_ms_per_step": v[4] / reps,
                               "tflops": v[1] / (v[3] * 1e-3) / 1e12, "peak_tflops": v[5] / 1e12,
                               "frac_of_issued_peak": v[1] / (v[3] * 1e-3) / v[5]} if not k.endswith("[valu]") else
                              # vector-ALU first-layer kernels (conv_smallc.hip): one pass over a C_out x L tensor
                              {"launches_per_step": v[0] // reps, "ms_per_step": v[3] / reps, "bound": "hbm",
                               "algorithmic_TBps": v[2] / (v[3] * 1e-3) / 1e12, "peak_TBps": HBM_PEAK / 1e12,
                               "frac_of_hbm_peak": v[2] / (v[3] * 1e-3) / HBM_PEAK}) for k, v in byk.items()}
        out["roofline"]["mfma_note"] = ("frac is against the NOMINAL rate of the issued MFMA (2.4 GHz); under a pure "
                                        "16-bit MFMA load with random operands the part is power-limited: the shader "
                                        "clock reads ~1.6 GHz (s_memtime / wall) and one MFMA takes 17.9 (bf16) / 19.5 (f16) ns per "
                                        "SIMD with two waves = 0.68-0.75 of nominal (tools/probe/f16x3.hip); up to 2 independent VALU per MFMA gap cost nothing in wall "
                                        "time, each further one ~1.2 ns (tools/probe/mfma_clock.hip, "
                                        "profiles/round4_probe_mfma_clock.txt).  frac_vs_sustained_mfma_rate prices the kernel "
                                        "against that measured rate")
        out["kernel_families"] = {k: {"launches_per_step": v[0] // reps, "ms_per_step": v[3] / reps, "call_ms_per_step": v[4] / reps,
                                      "tflops": v[1] / (v[3] * 1e-3) / 1e12,
                                      "algorithmic_GBps": v[2] / (v[3] * 1e-3) / 1e9} for k, v in agg.items()}
        # ---- forward-only leg of the north-star target (PQMF + conv stacks, no_grad)
        def fwd_once(force_prep):
            if force_prep:
                m._prep[0].invalidate()  # as if a parameter had changed: weight norm + repack of all layers
            m.prepare_weights(reuse=True)   # unchanged parameters: the packed operands are reused (rave_amd.prep.WeightPrep.run)
            m.decode(m.encoder.reparametrize(m.encode(x))[0])
            m.release_weights()

        def time_fwd(force_prep, nf=8):
            for _ in range(2):
                fwd_once(force_prep)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(nf):
                fwd_once(force_prep)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / nf

        def time_fwd_graph(nf=20):
            """The same forward recorded once into a hipGraph and replayed (the eager loop above pays ~50 Python-side launches per
            pass, which the GPU side of this leg -- 52 kernels of 20-60 us -- does not always hide)."""
            from rave_amd import ops as R
            m.prepare_weights(reuse=True)
            m.set_phase_flags_eagerly()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                R.range_reset(x.device)
                yy = m.decode(m.encoder.reparametrize(m.encode(x))[0])
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(nf):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            assert bool(torch.isfinite(yy).all())
            m.release_weights()
            return e0.elapsed_time(e1) * 1e-3 / nf

        with torch.no_grad():
            m.prepare_weights(reuse=True)
            t_fwd_prep = time_fwd(True)
            t_fwd_eager = t_fwd = time_fwd(False)
            fwd_mode = "eager"
            if use_graph:
                try:
                    t_fwd_graph = time_fwd_graph()
                    if t_fwd_graph < t_fwd:
                        t_fwd, fwd_mode = t_fwd_graph, "hipGraph replay"
                except Exception as e:                  # noqa: BLE001 -- secondary leg
                    fwd_mode = f"eager (forward capture failed: {type(e).__name__}: {str(e)[:120]})"
        fb = args.batch * V2_FWD_ACT_BYTES_PER_CLIP * (args.n_signal / 65536) + V2_WEIGHT_BYTES
        ff = 2.0 * args.batch * V2_FWD_MAC_PER_CLIP * (args.n_signal / 65536)
        out["forward_only"] = {"ms": t_fwd * 1e3, "mode": fwd_mode, "ms_eager": t_fwd_eager * 1e3,
                               "ms_with_weight_prep": t_fwd_prep * 1e3,
                               "note": "inference forward (no_grad): packed weights reused while no parameter changed; "
                                       "ms_with_weight_prep = the same with weight norm + repack of all 56 layers forced per call",
                               "algorithmic_bytes": fb, "algorithmic_flop": ff,
                               "hbm_roofline_frac": fb / t_fwd / HBM_PEAK,
                               "f32_mfma_frac": ff / t_fwd / F32_MFMA_PEAK,
                               "x6_mfma_frac": ff / t_fwd / X6_MFMA_PEAK,
                               "samples_per_s": args.batch * args.n_signal / t_fwd}
    if rank == 0 and world == 1 and not args.no_kernel_timing and not args.no_products_leg and args.config == "v2" \
            and args.batch == 32 and args.n_signal == 65536:
        out.update(bf16x6_leg(args.batch, args.n_signal))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "v2":
        out["cpu_baseline"] = cpu_baseline(args.n_signal)
    if use_ddp:
        tot = red_gen.bytes_reduced + (red_dis.bytes_reduced if red_dis else 0)
        ovl = red_gen.bytes_overlapped + (red_dis.bytes_overlapped if red_dis else 0)
        nst = max(args.warmup + args.steps, 1)
        exp_ms = sorted(a.elapsed_time(b) for a, b in ddp_exposed) if ddp_exposed else []
        # bytes one generator step all-reduces = the flat buckets of the stepping optimizer (a static figure: under graph
        # replay the Python-side counters only see the warm-up / capture / instrumented eager steps)
        per_step = sum(b.numel for b in red_gen.buckets) * 4
        out["ddp"] = {"allreduce_bytes_per_step": per_step, "buckets": len(red_gen.buckets),
                      "bucket_mib": [round(b.numel * 4 / 2 ** 20, 2) for b in red_gen.buckets], "backend": "nccl (RCCL)",
                      "exposed_ms": exp_ms[len(exp_ms) // 2] if exp_ms else None,
                      "exposed_ms_note": "median GPU time between the end of backward and the last bucket's all-reduce on "
                                         "the compute stream (HIP events around GradReducer.finish() in eager steps)",
                      "bytes_packed_total": red_gen.bytes_packed + (red_dis.bytes_packed if red_dis else 0),
                      "bytes_packed_per_step": (red_gen.bytes_packed + (red_dis.bytes_packed if red_dis else 0)) // nst,
                      "bytes_reduced_total": tot, "bytes_overlapped_total": ovl,
                      "allreduces_issued_from_side_stream": red_gen.side_issued + (red_dis.side_issued if red_dis else 0),
                      "step_mode": out["step_mode"],
                      "phase_note": ("every N runs the SAME per-GPU workload (BASELINE configs[1]: VAE-phase step, 32 clips per GPU) so "
                                     "that the scaling curve compares like with like; configs[2]'s VAE+GAN alternation: --phase gan"),
                      "overlap_fraction": ovl / tot if tot else None,
                      "overlap_note": "share of the all-reduce bytes issued from a gradient hook, i.e. while backward still ran "
                                      "(counted over the steps that ran Python: warm-up, capture, instrumented eager steps)",
                      "buffer_broadcast_bytes_per_step": bufsync.bytes_per_sync,
                      "buffer_bytes_static_not_resent": bufsync.bytes_static}
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
