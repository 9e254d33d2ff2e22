"""world_size-2 gloo test (CPU) of the data-parallel gradient reducer (rave_amd/ddp.py): bucketed
all-reduce launched from post-accumulate-grad hooks must equal the full-batch gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv1d(4, 8, 3, padding=1), nn.LeakyReLU(0.2), nn.Conv1d(8, 8, 3, padding=1),
                         nn.LeakyReLU(0.2), nn.Conv1d(8, 2, 1))


def _worker(rank, world, port, overlap, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rave_amd.ddp import GradReducer, broadcast_module, shard_batch
    net = _make_net()
    if rank == 1:  # perturb rank 1, broadcast must restore rank-0 weights
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    broadcast_module(net)
    extra = nn.Parameter(torch.zeros(3))          # a parameter that never receives a gradient
    red = GradReducer(list(net.parameters()) + [extra], bucket_mb=0.0005, overlap=overlap)
    assert len(red.buckets) > 1
    torch.manual_seed(1)
    x = torch.randn(8, 4, 32)
    per = shard_batch(8, rank, world)
    xs = x[rank * per:(rank + 1) * per]
    for step in range(2):
        net.zero_grad()
        red.begin()
        loss = net(xs).pow(2).mean()
        loss.backward()
        red.finish()
    torch.save((rank, [p.grad.clone() for p in net.parameters()], extra.grad),
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run(overlap, outdir):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, str(outdir))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    out = [torch.load(os.path.join(str(outdir), f"rank{r}.pt"), weights_only=False) for r in range(2)]
    net = _make_net()
    torch.manual_seed(1)
    x = torch.randn(8, 4, 32)
    # mean over ranks of per-shard mean losses == full-batch mean loss (equal shards)
    net(x).pow(2).mean().backward()
    for rank, grads, extra in out:
        assert extra is None
        for g, p in zip(grads, net.parameters()):
            assert torch.allclose(g, p.grad, rtol=1e-5, atol=1e-7)


def test_grad_reducer_overlapped(tmp_path):
    _run(True, tmp_path)


def test_grad_reducer_flush_only(tmp_path):
    _run(False, tmp_path)


class _EmaNet(nn.Module):
    """A buffer updated from LOCAL batch statistics (the RVQ codebooks / BatchNorm running stats pattern)."""

    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(4, 4)
        self.register_buffer("ema", torch.zeros(4))
        self.register_buffer("count", torch.zeros((), dtype=torch.long))

    def forward(self, x):
        if self.training:
            with torch.no_grad():
                self.ema.mul_(0.9).add_(x.mean(0), alpha=0.1)
                self.count += 1
        return self.lin(x - self.ema)


def _buf_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rave_amd.ddp import BufferSync
    torch.manual_seed(0)
    net = _EmaNet()
    sync = BufferSync(net)
    torch.manual_seed(10 + rank)                 # every rank sees different data
    for step in range(3):
        sync.sync()                              # start of step: rank-0 buffers everywhere
        net(torch.randn(16, 4) + rank)
    sync.sync()
    torch.save((net.ema.clone(), net.count.clone(), sync.bytes_sent), os.path.join(outdir, f"buf{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_buffer_sync_keeps_ranks_on_rank0_statistics(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_buf_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    e0, c0, sent = torch.load(os.path.join(str(tmp_path), "buf0.pt"), weights_only=False)
    e1, c1, _ = torch.load(os.path.join(str(tmp_path), "buf1.pt"), weights_only=False)
    assert torch.equal(e0, e1) and torch.equal(c0, c1) and int(c0) == 3 and sent == 4 * 4 * 4
    # rank 0 alone (torch DDP semantics: its buffers win)
    torch.manual_seed(0)
    ref = _EmaNet()
    torch.manual_seed(10)
    for step in range(3):
        ref(torch.randn(16, 4))
    assert torch.equal(ref.ema, e0)


class _SlotLinearFn(torch.autograd.Function):
    """y = x @ w.T whose backward writes dW where rave_amd.ops._grad_out says -- the protocol the HIP conv operators
    follow (rave_amd/ops.py): a parameter's bucket view while the reducer marked it fresh, else a new tensor."""

    @staticmethod
    def forward(ctx, x, w):
        from rave_amd.ops import _slot_of
        ctx.slot = _slot_of(w)
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from rave_amd.ops import _grad_out
        x, w = ctx.saved_tensors
        dw = _grad_out(ctx.slot, w.shape, w.device)
        torch.mm(dy.t(), x, out=dw)
        return dy @ w, dw


def _slot_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rave_amd.ddp import GradReducer
    torch.manual_seed(0)
    w1 = nn.Parameter(torch.randn(6, 4))      # gradient written in place by the operator: adopted, never copied
    w2 = nn.Parameter(torch.randn(3, 6))      # gradient produced by a torch op: copied into its view by the hook
    w3 = nn.Parameter(torch.randn(6, 4))      # shared weight (two uses): autograd sums the two contributions -> one copy
    red = GradReducer([w1, w2, w3], bucket_mb=1e-5)
    torch.manual_seed(5)
    x = torch.randn(8, 4)
    xs = x[rank * 4:(rank + 1) * 4]
    res = []
    for step in range(3):
        for p in (w1, w2, w3):
            p.grad = None
        red.begin()
        h = _SlotLinearFn.apply(xs, w1) + _SlotLinearFn.apply(xs, w3) + 0.5 * _SlotLinearFn.apply(xs * 2, w3)
        (h @ w2.t()).pow(2).mean().backward()
        red.finish()
        inside = all(any(b.flat.data_ptr() <= p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4
                         for b in red.buckets) for p in (w1, w2, w3))
        res.append((w1.grad.clone(), w2.grad.clone(), w3.grad.clone(), inside))
    torch.save((res, red.bytes_packed), os.path.join(outdir, f"slot{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradients_live_in_the_buckets_and_in_place_writers_skip_the_pack(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_slot_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    out = [torch.load(os.path.join(str(tmp_path), f"slot{r}.pt"), weights_only=False) for r in range(2)]
    # reference: full batch, plain autograd
    torch.manual_seed(0)
    w1 = torch.randn(6, 4, requires_grad=True)
    w2 = torch.randn(3, 6, requires_grad=True)
    w3 = torch.randn(6, 4, requires_grad=True)
    torch.manual_seed(5)
    x = torch.randn(8, 4)
    h = x @ w1.t() + x @ w3.t() + 0.5 * ((x * 2) @ w3.t())
    (h @ w2.t()).pow(2).mean().backward()
    for res, packed in out:
        for g1, g2, g3, inside in res:
            assert inside                                       # p.grad is a view into a flat all-reduce bucket
            assert torch.allclose(g1, w1.grad, rtol=1e-5, atol=1e-6)
            assert torch.allclose(g2, w2.grad, rtol=1e-5, atol=1e-6)
            assert torch.allclose(g3, w3.grad, rtol=1e-5, atol=1e-6)
        # w1 (written in place by its operator) was never copied; the torch-op and the shared-weight gradients once per step
        assert packed == 3 * (w2.numel() + w3.numel()) * 4
    for (a, _), (b, _) in [(out[0], out[1])]:
        for ra, rb in zip(a, b):
            assert all(torch.equal(u, v) for u, v in zip(ra[:3], rb[:3]))


def test_last_bucket_is_cut_short():
    """Bucket layout of GradReducer (no process group needed): reverse registration order, ~bucket_mb per bucket, and the
    LAST bucket -- whose all-reduce completes with the last gradient of backward and overlaps nothing -- at most tail_mb."""
    from rave_amd.ddp import GradReducer
    sizes = [10, 20, 300_000, 5, 4_000_000, 9_000_000, 100, 2_000_000, 50]
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    red = GradReducer(ps, bucket_mb=32.0, tail_mb=4.0)
    flat_order = [p for b in red.buckets for p in b.params]
    assert [p.numel() for p in flat_order] == sizes[::-1]                    # reverse registration order, nothing lost
    mib = [b.numel * 4 / 2 ** 20 for b in red.buckets]
    assert len(red.buckets) == 3 and mib[-1] <= 4.0 and mib[0] >= 32.0
    assert [len(b.params) for b in red.buckets] == [4, 1, 4]
    # a single small bucket is left alone; tail_mb = 0 disables the cut
    assert len(GradReducer(ps[:2], bucket_mb=8.0).buckets) == 1
    assert len(GradReducer(ps, bucket_mb=32.0, tail_mb=0.0).buckets) == 2
    for p in ps:                                                             # every parameter got its view
        v, fresh, active = p._rh_grad_slot
        assert v.shape == p.shape and fresh is False and active is False
    red.remove()


def _worker8(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from rave_amd.ddp import BufferSync, GradReducer, broadcast_module, shard_batch
    net = _make_net()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.1 * rank)                    # every rank starts elsewhere: the broadcast must bring rank 0's weights
    broadcast_module(net)
    red = GradReducer(list(net.parameters()), bucket_mb=0.0005, tail_mb=0.0002)
    per = shard_batch(256, rank, world)           # BASELINE configs[2]: global batch 256 -> 32 clips per GPU
    assert per == 32
    torch.manual_seed(1)
    x = torch.randn(256, 4, 16)
    xs = x[rank * per:(rank + 1) * per]
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        red.begin()
        net(xs).pow(2).mean().backward()
        red.finish()
        if step == 0:
            g0 = [p.grad.clone() for p in net.parameters()]
        opt.step()
    torch.save((rank, g0, [p.detach().clone() for p in net.parameters()], len(red.buckets), red.bytes_reduced,
                red.bytes_overlapped), os.path.join(outdir, f"w8_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_8_partition_and_buckets(tmp_path):
    """the N = 8 partition / bucket logic executed at least once (gloo, CPU): shard_batch(256, r, 8) = 32
    clips per rank, the bucketed all-reduce of 8 ranks equals the gradient of the full batch of 256, and two optimizer
    steps leave every rank with identical parameters."""
    world = 8
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    out = [torch.load(os.path.join(str(tmp_path), f"w8_{r}.pt"), weights_only=False) for r in range(world)]
    net = _make_net()
    torch.manual_seed(1)
    x = torch.randn(256, 4, 16)
    net(x).pow(2).mean().backward()
    for rank, g0, params, n_buckets, reduced, overlapped in out:
        assert n_buckets >= 2 and reduced > 0 and overlapped > 0
        for g, p in zip(g0, net.parameters()):
            assert torch.allclose(g, p.grad, rtol=1e-5, atol=1e-7)
        for a, b in zip(params, out[0][2]):
            assert torch.equal(a, b)             # every rank holds rank 0's parameters after two steps
    from rave_amd.ddp import shard_batch
    import pytest
    with pytest.raises(ValueError):
        shard_batch(100, 0, 8)


def test_buffer_sync_skips_buffers_no_training_step_writes():
    """BufferSync re-sends only state that can change between steps: RVQ codebooks and flags travel; PQMF
    filter banks, STFT windows and AdaIN's statistics (identity in training mode, rave/blocks.py:901-902) are identical on
    every rank after broadcast_module() and stay behind.  ``all_buffers=True`` = torch DDP's broadcast_buffers."""
    from rave_amd import model as M
    from rave_amd.ddp import BufferSync
    m = M.build_v3(capacity=4, latent_size=4)
    s = BufferSync(m)
    names = {id(b): k for k, b in m.named_buffers()}
    sent = {names[id(b)] for b in s.fbufs + s.ibufs}
    skipped = {names[id(b)] for b in s.skipped}
    assert "pqmf.hk" in skipped and "pqmf.h" in skipped
    assert any(k.endswith(".mean_x") for k in skipped) and any(k.endswith("window") or ".window_" in k for k in skipped)
    assert "receptive_field" in sent and "encoder.warmed_up" in sent
    assert not any(k.endswith((".mean_x", ".std_y")) or k.startswith("pqmf.") for k in sent)
    full = BufferSync(m, all_buffers=True)
    assert not full.skipped and full.bytes_per_sync == s.bytes_per_sync + s.bytes_static
    d = M.build_discrete(capacity=4, latent_size=4, num_quantizers=2, codebook_size=8, noise_augmentation=2)
    sd_ = BufferSync(d)
    dn = {id(b): k for k, b in d.named_buffers()}
    sent_d = {dn[id(b)] for b in sd_.fbufs + sd_.ibufs}
    for k in ("encoder.rvq.layers.0._codebook.embed", "encoder.rvq.layers.1._codebook.cluster_size",
              "encoder.rvq.layers.0._codebook.inited", "encoder.enabled"):
        assert k in sent_d, k
