"""Two data-parallel ranks driving the REAL HIP training step on one GPU (-m gpu).

RCCL refuses two ranks on one device ("Duplicate GPU detected", probed: tools/debug/nccl2_one_gpu.py) and a gpurun
box has a single MI355X, so the two ranks exchange their CUDA gradient buckets through the gloo backend here; the
RCCL calls themselves are exercised by bench.py with RAVE_FORCE_DIST=1.  Everything else is the production path:
HIP forward / backward, post-accumulate-grad hooks firing DURING backward, asynchronous bucket all-reduce, averaging
before Adam, rank-0 buffer broadcast before the forward."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev):
    from rave_amd import model as M
    torch.manual_seed(0)
    return M.build_v2(capacity=16, latent_size=16).to(dev).train()


def _data():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import rave_oracle as O
    x = O.synthetic_batch(4, 1, 32768, seed=9)
    eps = torch.randn(4, 16, 16, generator=torch.Generator().manual_seed(3))
    return x, eps


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rave_amd import ddp
    m = _model(dev)
    if rank == 1:                       # perturbed replica: the start-up broadcast must restore rank 0's weights
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
            m.receptive_field.fill_(7)
    ddp.broadcast_module(m)
    m.configure_optimizers()
    gen = list(m.encoder.parameters()) + list(m.decoder.parameters())
    red = ddp.GradReducer(gen, bucket_mb=0.25)
    sync = ddp.BufferSync(m)
    x, eps = _data()
    per = ddp.shard_batch(4, rank, world)
    sl = slice(rank * per, (rank + 1) * per)
    sync.sync()
    for it in range(2):                 # the second step starts from adopted bucket views (zero_grad -> None -> views again)
        m.training_step(x[sl].to(dev), 0, eps=eps[sl].to(dev), grad_begin=lambda idx: red.begin(),
                        grad_sync=lambda idx: red.finish())
        if it == 0:                     # compare the FIRST step's gradients (the reference below runs one step)
            torch.cuda.synchronize()
            grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    flat_ranges = [(b.flat.data_ptr(), b.flat.data_ptr() + b.flat.numel() * 4) for b in red.buckets]
    in_bucket = sum(1 for p in gen if p.grad is not None and any(lo <= p.grad.data_ptr() < hi for lo, hi in flat_ranges))
    torch.save(dict(grads=grads, buckets=len(red.buckets), reduced=red.bytes_reduced, overlapped=red.bytes_overlapped,
                    packed=red.bytes_packed, in_bucket=in_bucket, n_gen=sum(1 for p in gen if p.grad is not None),
                    rf=int(m.receptive_field.sum())), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_average_the_hip_gradients_and_overlap(tmp_path):
    dev = torch.device("cuda", 0)
    # reference: the two shards one after the other in this process, gradients averaged by hand
    x, eps = _data()
    want = None
    for r in range(2):
        m = _model(dev)
        m.configure_optimizers()
        m.training_step(x[2 * r:2 * r + 2].to(dev), 0, eps=eps[2 * r:2 * r + 2].to(dev))
        g = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
        want = g if want is None else {k: 0.5 * (want[k] + g[k]) for k in g}
        del m
    torch.cuda.synchronize()
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    outs = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=False) for r in range(2)]
    for o in outs:
        assert o["buckets"] > 1 and o["reduced"] > 0
        assert o["overlapped"] > 0                      # at least one bucket left from a hook, i.e. during backward
        assert o["rf"] == 0                             # rank-0 buffers won
        assert o["in_bucket"] == o["n_gen"] > 50        # every gradient LIVES in its all-reduce bucket (no copy-back)
        assert o["packed"] == 0                         # ... and was written there by the backward kernels (no pack pass)
        assert set(o["grads"]) == set(want)
        for k, g in o["grads"].items():
            err = float((g.double() - want[k].double()).norm() / (want[k].double().norm() + 1e-30))
            assert err < 1e-5, (k, err)
    for k in outs[0]["grads"]:
        assert torch.equal(outs[0]["grads"][k], outs[1]["grads"][k])
