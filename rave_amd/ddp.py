"""Data-parallel gradient averaging for one process per GPU (RCCL over xGMI through
``torch.distributed`` backend "nccl"; "gloo" on CPU for tests).

The reference has no collective call of its own: multi-GPU training is Lightning's implicit DDP
(scripts/train.py:242-255).  This is the MI355X-native equivalent for the hot
path: clips are independent, so the only exchange is the gradient mean.

* parameters are grouped into ~32 MiB buckets in REVERSE registration order (= the order their
  gradients become final during backward: discriminator -> decoder -> encoder), the last bucket cut down to <= 4 MiB
  (its all-reduce is the only one nothing can hide); every bucket owns ONE persistent flat buffer and every parameter a
  view into it;
* **gradients live in the buckets**: the weight-gradient / weight-norm backward kernels of the conv operators write
  straight into the parameter's view (rave_amd.ops: ``grad_slot``) and hand that view to autograd, which adopts it as
  ``p.grad`` -- no pack pass before the all-reduce and no copy-back after it (round 2 paid 2 x 126 MB of copies per
  step).  A gradient produced by anything else (torch ops: Snake alpha, BatchNorm) is copied into its view once by
  the hook and ``p.grad`` re-bound to the view;
* a post-accumulate-grad hook per parameter counts arrivals; when a bucket is complete ONE asynchronous all-reduce
  over its flat buffer is enqueued (RCCL's own stream: it overlaps the remaining backward kernels).  The reduction
  is ``AVG`` where the backend has it (RCCL), else ``SUM`` + one scale pass (gloo);
* ``finish()`` (between backward and optimizer.step) waits.  Buckets that did not complete (parameters without a
  gradient this step) are flushed with the missing views zeroed; their ``p.grad`` stays ``None``;
* everything is stream-ordered (no host synchronisation): ``begin() ... backward ... finish()`` can be recorded into a
  hipGraph together with the rest of the step -- RCCL collectives are capturable -- so a data-parallel rank replays
  the same single graph launch per step as a single-GPU run (rave_amd.model.GraphedTrainingStep(grad_sync=...)).

xGMI is point-to-point (7 links per GPU); a few large messages keep every link busy, hence the
larger-than-NCCL-default bucket.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat: Optional[torch.Tensor] = None
        self.views: List[torch.Tensor] = []
        self.pending = 0
        self.arrived: List[bool] = []
        self.index = {}
        self.work = None
        self.launched = False


class GradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 32.0,
                 process_group=None, overlap: bool = True, force: bool = False, tail_mb: float = 4.0):
        self.pg = process_group
        self.force = force   # run the collectives even with a single rank (smoke-testing the RCCL path)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        plist = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * 1024 * 1024 / 4)
        groups: List[List[torch.nn.Parameter]] = []
        cur: List[torch.nn.Parameter] = []
        n = 0
        for p in reversed(plist):
            cur.append(p)
            n += p.numel()
            if n >= cap:
                groups.append(cur)
                cur, n = [], 0
        if cur:
            groups.append(cur)
        # The LAST bucket's all-reduce cannot overlap anything (it completes with the last gradient of backward), so it
        # is kept small: the trailing parameters (the first layers of the network, registered first) are peeled off into
        # a bucket of at most `tail_mb`; whatever else shared their bucket leaves as soon as it is complete.
        tcap = int(tail_mb * 1024 * 1024 / 4)
        if groups and tcap > 0 and sum(p.numel() for p in groups[-1]) > tcap and len(groups[-1]) > 1:
            last = groups[-1]
            k, acc = len(last), 0
            while k > 1 and acc + last[k - 1].numel() <= tcap:
                k -= 1
                acc += last[k].numel()
            if 0 < k < len(last):
                groups[-1:] = [last[:k], last[k:]]
        self.buckets: List[_Bucket] = [_Bucket(g) for g in groups]
        self._where = {}
        self._hooks = []
        for b in self.buckets:
            p0 = b.params[0]
            b.flat = torch.zeros(b.numel, dtype=p0.dtype, device=p0.device)
            o = 0
            for i, p in enumerate(b.params):
                if p.dtype != p0.dtype or p.device != p0.device:
                    raise ValueError("GradReducer: parameters of one reducer must share dtype and device")
                v = b.flat[o:o + p.numel()].view(p.shape)
                o += p.numel()
                b.views.append(v)
                b.index[p] = i
                # [view, fresh]: the conv operators' backward writes the gradient into `view` and returns it while
                # `fresh` (set by begin(), cleared by the first writer of the step) -- rave_amd.ops.grad_slot
                p._rh_grad_slot = [v, False, False]      # [view, fresh, reducer active for this step]
                self._where[p] = b
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"          # RCCL reduces with ncclAvg; gloo has SUM only
        self.enabled = False
        self.bytes_reduced = 0
        self.bytes_overlapped = 0      # bytes whose all-reduce was issued from a hook, i.e. while backward still ran
        self.bytes_packed = 0          # bytes that had to be copied into a view (gradients not written in place)
        self.side_issued = 0           # all-reduces issued from the weight-gradient side stream (see _launch)
        self._in_finish = False

    # ---- lifecycle of one step
    def begin(self) -> None:
        """Call before backward (with the gradients of the step's parameters cleared: ``zero_grad(set_to_none=True)``)."""
        self.enabled = self.world > 1 or self.force
        for b in self.buckets:
            b.pending = len(b.params)
            b.arrived = [False] * len(b.params)
            b.work = None
            b.launched = False
            for p in b.params:
                p._rh_grad_slot[1] = self.enabled and p.grad is None
                p._rh_grad_slot[2] = self.enabled

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self.enabled:
            return
        b = self._where[p]
        i = b.index[p]
        v = b.views[i]
        g = p.grad
        if g is not None and g.data_ptr() != v.data_ptr():
            with torch.no_grad():
                v.copy_(g)             # produced outside the conv operators: one copy, then the view IS the gradient
            p.grad = v
            self.bytes_packed += p.numel() * 4
        if not b.arrived[i]:
            b.arrived[i] = True
            b.pending -= 1
        if b.pending == 0 and self.overlap and not b.launched:
            self._launch(b)

    def _launch(self, b: _Bucket) -> None:
        missing = [v for v, a in zip(b.views, b.arrived) if not a]
        if len(missing) == len(b.views):
            return                     # nothing in this bucket received a gradient this step
        if missing:
            with torch.no_grad():
                torch._foreach_zero_(missing)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        side = None
        if b.flat.is_cuda:
            # Gradients of this bucket may still be in flight on the weight-gradient side stream (rave_amd.ops._OnSide; the
            # collected weight-norm backward of the branch is enqueued there now).  The collective is issued FROM that stream,
            # which first waits for the compute stream: it is ordered behind every writer of the bucket on both streams, and
            # the compute stream -- the data-gradient chain the rest of backward waits for -- is not stalled at the bucket
            # boundary.  Nothing in flight there (side stream off, bucket of torch-op gradients): issued from the compute stream.
            from . import ops
            side = ops.side_stream_for_collective(b.flat.device)
        if side is not None:
            with torch.cuda.stream(side):
                b.work = dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True)
            self.side_issued += 1
        else:
            b.work = dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True)
        b.launched = True
        self.bytes_reduced += b.numel * 4
        if not self._in_finish:
            self.bytes_overlapped += b.numel * 4

    def finish(self) -> None:
        """Call after backward, before optimizer.step(): flush incomplete buckets, wait for the collectives."""
        if not self.enabled:
            return
        self._in_finish = True
        for b in self.buckets:
            if not b.launched:
                self._launch(b)      # incomplete bucket or overlap disabled
        self._in_finish = False
        for b in self.buckets:
            if b.work is None:
                continue
            b.work.wait()
            if not self._avg:
                with torch.no_grad():
                    b.flat.mul_(1.0 / self.world)
            b.work = None
        for b in self.buckets:
            for p in b.params:
                p._rh_grad_slot[1] = False
                p._rh_grad_slot[2] = False
        self.enabled = False

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for b in self.buckets:
            for p in b.params:
                if hasattr(p, "_rh_grad_slot"):
                    del p._rh_grad_slot


class BufferSync:
    """Rank-0 module buffers to every rank before each forward -- what the reference gets from torch DDP's
    ``broadcast_buffers=True`` under Lightning (scripts/train.py:242-255): the RVQ codebooks
    (``embed / embed_avg / cluster_size / inited``, rave/quantization.py:97-100,165-179 -- "buffers are in sync
    and all the workers will take the same decision") and the v1 encoder's BatchNorm running statistics are
    updated from LOCAL batch statistics on every rank and would drift apart without it.

    All floating-point buffers travel as ONE flat message per step (17 MB for the 16-stage RVQ; xGMI is
    point-to-point, one large broadcast beats hundreds of small ones); integer buffers (``receptive_field``,
    BatchNorm's ``num_batches_tracked``) are broadcast one by one."""

    # Buffers that no training step writes: filter-bank coefficients, STFT windows, and AdaIN's statistics (the module is
    # the identity in training mode, rave/blocks.py:901-902; its buffers only move in eval mode).  They are identical on every
    # rank after broadcast_module() and stay so: not re-sent every step (round 4 broadcast all of them: 1.5 MB of PQMF /
    # window constants and 22 x 4 AdaIN tensors per v3 step).
    STATIC_OWNERS = ("CachedPQMF", "PQMF", "MultiScaleSTFT", "AdaptiveInstanceNormalization", "Spectrogram", "_Stft")

    def __init__(self, module: torch.nn.Module, src: int = 0, process_group=None, force: bool = False,
                 all_buffers: bool = False):
        """``all_buffers=True``: torch DDP's exact ``broadcast_buffers`` behaviour (every buffer, every step)."""
        self.pg, self.src, self.force = process_group, src, force
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        bufs, self.skipped = [], []
        seen = set()
        for mod in module.modules():
            static = (not all_buffers) and type(mod).__name__ in self.STATIC_OWNERS
            for name, b in mod.named_buffers(recurse=False):
                if b.numel() == 0 or id(b) in seen:
                    continue
                seen.add(id(b))
                # (a static owner's own flags / counters are tiny and may be host-visible state: only its float tables stay)
                if static and b.is_floating_point():
                    self.skipped.append(b)
                else:
                    bufs.append(b)
        self.fbufs = [b for b in bufs if b.is_floating_point()]
        self.ibufs = [b for b in bufs if not b.is_floating_point()]
        self.flat: Optional[torch.Tensor] = None
        self.bytes_sent = 0
        self.bytes_per_sync = sum(b.numel() * b.element_size() for b in bufs)
        self.bytes_static = sum(b.numel() * b.element_size() for b in self.skipped)

    def sync(self) -> None:
        """Call at the start of every step (before the forward pass)."""
        if not (self.world > 1 or (self.force and dist.is_initialized())):
            return
        with torch.no_grad():
            groups = {}
            for b in self.fbufs:
                groups.setdefault((b.dtype, b.device), []).append(b)
            for (dtype, device), bs in groups.items():
                n = sum(b.numel() for b in bs)
                flat = torch.empty(n, dtype=dtype, device=device)
                views, o = [], 0
                for b in bs:
                    views.append(flat[o:o + b.numel()].view_as(b))
                    o += b.numel()
                torch._foreach_copy_(views, [b.data for b in bs])
                dist.broadcast(flat, src=self.src, group=self.pg)
                torch._foreach_copy_([b.data for b in bs], views)
                self.bytes_sent += n * flat.element_size()
            for b in self.ibufs:
                dist.broadcast(b.data, src=self.src, group=self.pg)


def broadcast_module(module: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Rank-0 parameters AND buffers to every rank (what torch DDP does at construction and, for
    buffers, before each forward: SURVEY.md section 2.3 C2)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=process_group)


def shard_batch(global_batch: int, rank: int, world: int) -> int:
    """Even split of the audio minibatch across ranks: 256 -> 32 per GPU."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    return global_batch // world
