"""Worker of tests/test_gpu_dispatch.py::test_graph_replay_at_the_benchmarked_size_* (run as a script, one process per
case, because the data-parallel variant owns a process group):

the BENCHMARKED execution mode of bench.py -- v2 at CAPACITY 96, batch 32 x 65536, hipGraph replay, the weight-gradient
branch on its second stream (default), optionally the data-parallel machinery of one rank forced on (``RAVE_FORCE_DIST=1``:
bucket views, RCCL all-reduce and buffer broadcast recorded into the same graph) -- against the eager step:

  1. ``steps`` eager steps vs ``steps`` replayed steps from the same seeds: every parameter BIT-identical;
  2. ``replays`` replays of ONE captured step, each from the same restored parameters / optimizer state / input:
     every replay must leave bit-identical parameters (a race between the two branches of the graph would show as
     last-bit differences between replays: profiles/round3_negative_precompute_graph_race.txt).

Prints one JSON line.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rave_oracle as O  # noqa: E402  (the checker's synthetic batches only)


def main():
    steps = int(os.environ.get("GI_STEPS", "3"))
    replays = int(os.environ.get("GI_REPLAYS", "20"))
    batch = int(os.environ.get("GI_BATCH", "32"))
    capacity = int(os.environ.get("GI_CAPACITY", "96"))
    force_dist = os.environ.get("RAVE_FORCE_DIST", "0") == "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from rave_amd import ddp, model as M

    n_signal = 65536
    xs = [O.synthetic_batch(batch, 1, n_signal, seed=90 + i).to(dev) for i in range(steps + 1)]
    gen = torch.Generator().manual_seed(2)
    latent = 128 if capacity == 96 else 16
    es = [torch.randn(batch, latent, n_signal // 2048, generator=gen).to(dev) for _ in range(steps + 1)]

    def build():
        torch.manual_seed(0)
        m = (M.build_v2() if capacity == 96 else M.build_v2(capacity=capacity, latent_size=latent)).to(dev).train()
        m.configure_optimizers(capturable=True)
        kw, bufsync = {}, None
        if force_dist:
            red = ddp.GradReducer(list(m.encoder.parameters()) + list(m.decoder.parameters()), force=True)
            bufsync = ddp.BufferSync(m, force=True)
            kw = dict(grad_begin=lambda idx: red.begin(), grad_sync=lambda idx: red.finish())
        return m, kw, bufsync

    def params(m):
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in m.named_parameters()}

    out = {"dist": force_dist, "steps": steps, "replays": replays, "batch": batch, "capacity": capacity}

    # ---- 1. eager vs replay, `steps` steps
    m, kw, bufsync = build()
    for i in range(steps):
        if bufsync is not None:
            bufsync.sync()
        m.training_step(xs[i].clone(), i, eps=es[i], capture_safe=True, **kw)
        m.on_train_batch_end(None, None, i)
    pe = params(m)
    del m
    m, kw, bufsync = build()
    p0 = params(m)
    step = M.GraphedTrainingStep(m, xs[0], inject_eps=True, before_step=(bufsync.sync if bufsync is not None else None), **kw)
    for i in range(steps):
        step(xs[i], i, eps=es[i])
        m.on_train_batch_end(None, None, i)
    pg = params(m)
    out["eager_vs_graph_differing"] = [k for k in pe if not torch.equal(pe[k], pg[k])][:8]
    out["eager_vs_graph_n_differing"] = sum(1 for k in pe if not torch.equal(pe[k], pg[k]))
    out["params_moved"] = sum(1 for k in pe if not torch.equal(pe[k], p0[k]))
    out["n_params"] = len(pe)

    # ---- 2. `replays` replays of one captured step from the same state
    state = ({k: v.clone() for k, v in m.state_dict().items()}, [M._clone_opt(o) for o in m.optimizers()])
    first = None
    differing = []
    for r in range(replays):
        m.load_state_dict(state[0])
        for o, st in zip(m.optimizers(), state[1]):
            M._restore_opt(o, st)
        M._reset_host_shadows(m)
        step(xs[steps], steps, eps=es[steps])
        p = params(m)
        if first is None:
            first = p
        else:
            differing.append(sum(1 for k in p if not torch.equal(p[k], first[k])))
    out["replay_differing_tensors"] = differing
    out["replay_moved"] = sum(1 for k in first if not torch.equal(first[k], pg[k]))
    print("GRAPH_IDENTITY " + json.dumps(out))
    if force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
