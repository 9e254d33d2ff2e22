"""torch.autograd Functions over the C ABI (include/rave_hip.h).

Every function requires contiguous fp32 CUDA(ROCm) tensors and enqueues on the current stream.
There is no eager/PyTorch fallback: a missing library or a non-zero return code raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import ACT_LEAKY, ACT_NONE, ACT_SNAKE  # noqa: F401

Tensor = torch.Tensor


# ---- optional per-launch timing (bench.py roofline leg) -------------------------------------------------------------
# Two clocks per launch: (a) the MAIN kernel's own start / stop timestamps -- the library dispatches it with a pair of HIP
# events attached (rh_set_kernel_events -> hipExtLaunchKernelGGL), i.e. the duration rocprofv3 reports for that dispatch;
# (b) a torch event bracket on the launch stream around the whole C-ABI call (main kernel + its split-K finalize /
# partial-sum reduction launches + event overhead).
_PROFILE = None
_EVENT_POOL = []


def profile_begin() -> None:
    global _PROFILE
    _PROFILE = []


def _hip_event():
    if _EVENT_POOL:
        return _EVENT_POOL.pop()
    h = C.c_void_p()
    L.check(L.lib.rh_event_create(C.byref(h)), "event_create")
    return h


def profile_end():
    """[(kind, flops, bytes, kernel_ms, call_ms)] for every conv launch since profile_begin(): kernel_ms = the main kernel's
    own duration (None where the launch took a kernel without the hook), call_ms = the whole call on its stream."""
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    out = []
    for (k, f, b, e0, e1, h0, h1, used) in rec:
        kms = None
        if used:
            ms = C.c_float()
            if L.lib.rh_event_elapsed_ms(h0, h1, C.byref(ms)) == 0:
                kms = ms.value
        _EVENT_POOL.extend((h0, h1))
        out.append((k, f, b, kms, e0.elapsed_time(e1)))
    return out


def _timed(kind, f, b, fn):
    """Run one C-ABI launch under both clocks (profile mode only)."""
    h0, h1 = _hip_event(), _hip_event()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    L.lib.rh_set_kernel_events(h0, h1)
    r = fn()
    used = L.lib.rh_kernel_events_used()
    e1.record()
    _PROFILE.append((kind, f, b, e0, e1, h0, h1, used))
    return r


# ---- optional launch-plan log (parity tests: which template instance / K split / layout did each conv launch get?)
_PLAN_LOG = None


def plan_log_begin() -> None:
    global _PLAN_LOG
    _PLAN_LOG = []


def plan_log_end():
    """[(which, (c_in, c_out, kernel, stride, dilation, l_in), (family, tm, tn, wm, ksplit, input stride, vrows, workgroups))]"""
    global _PLAN_LOG
    rec, _PLAN_LOG = _PLAN_LOG, None
    return rec


def _log_plan(d, which: int, has_bias: bool, has_add: bool) -> None:
    if _PLAN_LOG is None:
        return
    out = (C.c_int32 * 8)()
    L.check(L.lib.rh_conv1d_plan_info(C.byref(d), which, int(has_bias), int(has_add), out), "conv1d_plan_info")
    _PLAN_LOG.append((which, (d.c_in, d.c_out, d.kernel, d.stride, d.dilation, d.l_in, d.transposed), tuple(out)))


# ---- optional shadow check (parity tests): every conv launch is repeated on the exact-f32 MFMA kernels (RH_CONV_X6=0 /
# RH_WGRAD_X6=0: bit-reproducible fmaf chains) with the SAME operands and the relative L2 difference recorded.  Per
# launch, so activation-gate flips cannot propagate from one layer into the comparison of the next.
_SHADOW = None


def shadow_check_begin() -> None:
    global _SHADOW
    _SHADOW = []


def shadow_check_end():
    """[(kind, (c_in, c_out, kernel, stride, dilation, l_in, transposed, batch), rel_l2)]"""
    global _SHADOW
    rec, _SHADOW = _SHADOW, None
    return rec


def _rel(a: Tensor, b: Tensor) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


class _ExactF32:
    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in ("RH_CONV_X6", "RH_WGRAD_X6")}
        os.environ["RH_CONV_X6"] = "0"
        os.environ["RH_WGRAD_X6"] = "0"

    def __exit__(self, *a):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _shadow(kind: str, d, run, outs) -> None:
    """``run(outs2)`` repeats the launch into fresh output tensors; called after the real launch."""
    if _SHADOW is None:
        return
    outs2 = [torch.empty_like(o) if o is not None else None for o in outs]
    with _ExactF32():
        check_rc = run(outs2)
    L.check(check_rc, "shadow " + kind)
    key = (d.c_in, d.c_out, d.kernel, d.stride, d.dilation, d.l_in, d.transposed, d.batch)
    for o, o2 in zip(outs, outs2):
        if o is not None:
            _SHADOW.append((kind, key, _rel(o, o2)))


# ---- optional gate log (parity tests): every conv launch in forward order with the parameters it carries and -- where a
# LeakyReLU is fused into its input staging -- the pre-activation tensor itself.  A LeakyReLU gate is the one discontinuity
# of the hot path's backward: a pre-activation within rounding of zero takes the other slope in another fp32 evaluation and
# changes that element's gradient 5x.  The tests COUNT such flips between two evaluations (sign masks of the logged tensors)
# and require every parameter gradient outside the tight bound to lie upstream (in the backward sense) of at least one.
_GATE_LOG = None


def gate_log_begin() -> None:
    global _GATE_LOG
    _GATE_LOG = []


def gate_log_end():
    """[(ids of the parameters of the launch, pre-activation tensor or None)] in forward launch order."""
    global _GATE_LOG
    rec, _GATE_LOG = _GATE_LOG, None
    return rec


def _log_gate(params, x, act: int) -> None:
    if _GATE_LOG is not None:
        _GATE_LOG.append((tuple(id(p) for p in params if p is not None), x.detach() if act == ACT_LEAKY else None))


def _conv_cost(d: "L.ConvDesc"):
    """Algorithmic FLOPs and bytes of one conv launch (SURVEY.md section 8d rule: input once, output
    once, weights once, fp32; activation / padding / residual count zero)."""
    pos = d.l_in if d.transposed else d.l_out * d.inner
    flops = 2.0 * d.batch * d.c_out * d.c_in * d.kernel * pos
    x_elems = d.batch * d.c_in * (d.in_valid if d.in_valid else d.l_in * d.inner)
    y_elems = d.batch * d.c_out * d.l_out * d.inner
    w_elems = d.c_out * d.c_in * d.kernel
    return flops, 4.0 * (x_elems + y_elems + w_elems)


# rh_conv1d_kernel_family: 0 f32-input MFMA, 1 bf16x6 MFMA, 2 vector-ALU first-layer kernels (conv_smallc.hip)
_FAMILY = {0: "[f32]", 1: "[x6]", 2: "[valu]"}


def _launch(kind: str, d, fn, has_bias=False, has_add=False):
    if _PROFILE is None:
        return fn()
    if kind in ("conv_fwd", "conv_dgrad"):   # which instruction does this launch issue?
        fam = L.lib.rh_conv1d_kernel_family(C.byref(d), 0 if kind == "conv_fwd" else 1, int(has_bias), int(has_add))
        kind += _FAMILY.get(fam, "[f32]")
    elif kind == "conv_wgrad":
        kind += _FAMILY.get(L.lib.rh_conv1d_bwd_weight_kernel_family(C.byref(d)), "[f32]")
    f, b = _conv_cost(d)
    return _timed(kind, f, b, fn)


def _chk(t: Optional[Tensor], name: str) -> Optional[Tensor]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"rave_amd: {name} must live on the GPU (the HIP hot path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"rave_amd: {name} must be float32, got {t.dtype}")
    return t.contiguous()


@dataclass(frozen=True)
class ConvGeom:
    """Static geometry of one convolution (see rh_conv1d_desc)."""
    stride: int = 1
    dilation: int = 1
    pad_left: int = 0
    pad_right: int = 0
    transposed: bool = False
    act: int = ACT_NONE
    slope: float = 0.2
    inner: int = 1          # W of a Conv2d with (k,1) kernel on (B,C,H,W)
    fold: bool = False      # input is the un-folded (B,C,T) waveform of MultiPeriodDiscriminator
    out_act: int = ACT_NONE  # LeakyReLU on the OUTPUT (descript WNConv2d + LeakyReLU(0.1))
    out_slope: float = 0.1

    def out_len(self, l_in: int, k: int) -> int:
        if self.transposed:
            return (l_in - 1) * self.stride - 2 * self.pad_left + k
        return (l_in + self.pad_left + self.pad_right - self.dilation * (k - 1) - 1) // self.stride + 1


def _desc(g: ConvGeom, batch, c_in, c_out, l_in, l_out, k, in_valid=0) -> L.ConvDesc:
    return L.ConvDesc(batch=batch, c_in=c_in, c_out=c_out, l_in=l_in, l_out=l_out, kernel=k,
                      stride=g.stride, dilation=g.dilation, pad_left=g.pad_left,
                      transposed=int(g.transposed), groups=1, inner=g.inner, in_valid=in_valid,
                      act=g.act, act_slope=g.slope, out_act=g.out_act, out_slope=g.out_slope)


def _ws(nbytes: int, device) -> Optional[Tensor]:
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32) if nbytes > 0 else None


# ---- range slots of the f16 matrix-core kernels (include/rave_hip.h: rh_x6_set_ranges; csrc/common.hpp: RH_X6_F16) ------------
# The x6 kernels scale every activation operand by a power of two derived from the TENSOR's max |x|.  That maximum travels with
# the tensor as a "range slot" (kRangeWords uint32 in device memory): the kernel that produces a tensor leaves it there
# (epilogue atomicMax: no extra pass), and the tensor OBJECT carries the slot as an attribute together with its version
# counter.  A tensor without a (current) slot -- produced by a torch op, a view, modified in place -- gets one from a pass of
# rh_amax_f32 the first time a convolution consumes it.  Slots come from a zeroed pool; RAVE.training_step takes a fresh pool per
# step (range_reset: one fill launch -- recorded into a hipGraph it re-zeroes the slots at every replay, so a replayed step
# computes the same scales as an eager one).
_RANGES = L.lib.rh_x6_uses_ranges() == 1
_RANGE_WORDS = L.lib.rh_x6_range_words()
_RANGE_SLOTS = 2048     # 4 KB each: more than any step of the shipped configs uses (a pool that runs out mid-step is replaced)
_RANGE_POOLS = {}        # device -> [pool, cursor, previous pool (kept alive: a side stream may still read it), stream of the reset]


def range_reset(device=None, _exhausted: bool = False) -> None:
    """Start a fresh zeroed slot pool (call at the start of a step; mandatory inside a hipGraph capture)."""
    if not _RANGES:
        return
    for dev in ([device] if device is not None else list(_RANGE_POOLS)):
        st = _RANGE_POOLS.get(dev)
        prev = st[0] if st else None
        cur = torch.cuda.current_stream(dev)
        pool = torch.zeros(_RANGE_SLOTS * _RANGE_WORDS, device=dev, dtype=torch.int32)
        if _exhausted:
            # a pool that ran out in the middle of a step, possibly on the weight-gradient side stream: the other stream of the
            # step must not publish into the new pool before its zero fill has run
            ev = torch.cuda.Event()
            ev.record(cur)
            d_ = dev if isinstance(dev, torch.device) else torch.device(dev)
            for other in [st[3] if st else None, _SIDE.get(d_.index if d_.index is not None else torch.cuda.current_device())]:
                if other is not None and other != cur:
                    other.wait_event(ev)
            main = st[3] if st else cur
        else:
            main = cur
        _RANGE_POOLS[dev] = [pool, 0, prev, main]


def _ranges_on() -> bool:
    return _RANGES and os.environ.get("RH_CONV_X6", "1") != "0"


def _new_range(device) -> Tensor:
    st = _RANGE_POOLS.get(device)
    if st is None or st[1] >= _RANGE_SLOTS:
        range_reset(device, _exhausted=st is not None)
        st = _RANGE_POOLS[device]
    i = st[1]
    st[1] = i + 1
    return st[0][i * _RANGE_WORDS:(i + 1) * _RANGE_WORDS]


def _attach_range(t: Optional[Tensor], slot: Optional[Tensor]) -> None:
    if t is not None and slot is not None:
        t._rh_range = (slot, t._version)


_RANGE_MISS = None       # diagnostics (range_miss_log_begin): [(tag, shape)] of the tensors that needed an rh_amax_f32 pass


def range_miss_log_begin() -> None:
    global _RANGE_MISS
    _RANGE_MISS = []


def range_miss_log_end():
    global _RANGE_MISS
    out, _RANGE_MISS = _RANGE_MISS, None
    return out


def _valid_slot(t: Tensor):
    r = getattr(t, "_rh_range", None)
    if r is not None and r[1] == t._version and r[0].device == t.device:
        return r[0]
    return None


def _range_of(t: Tensor, s, tag: str = "") -> Tensor:
    """The range slot of ``t``: the one its producer left -- also through a view that covers the whole producer tensor (same
    elements, same maximum; views share the version counter) --, else computed now (one pass over ``t`` on stream ``s``)."""
    slot = _valid_slot(t)
    if slot is not None:
        return slot
    base = t._base
    if base is not None and base.numel() == t.numel():
        slot = _valid_slot(base)
        if slot is not None:
            t._rh_range = (slot, t._version)
            return slot
    if _RANGE_MISS is not None:
        _RANGE_MISS.append((tag, tuple(t.shape)))
    slot = _new_range(t.device)
    L.check(L.lib.rh_amax_f32(L.ptr(t), t.numel(), L.ptr(slot), s), "amax")
    _attach_range(t, slot)
    return slot


def _fwd(d, x, wp, bias, alpha, residual, y, s):
    _log_plan(d, 0, bias is not None, residual is not None)
    ws = _ws(L.lib.rh_conv1d_fwd_workspace_bytes(C.byref(d)), y.device)
    rin = rout = None
    if _ranges_on():
        # the input's slot only where the f16 kernels will read it (family 1); the output's always (its consumers may)
        if L.lib.rh_conv1d_kernel_family(C.byref(d), 0, int(bias is not None), int(residual is not None)) == 1:
            rin = _range_of(x, s, "fwd x")
        rout = _new_range(y.device)

    def run(out, ranges=True):
        if ranges and rout is not None:
            L.lib.rh_x6_set_ranges(None, L.ptr(rin), L.ptr(rout), None)
        return L.lib.rh_conv1d_fwd_f32(C.byref(d), L.ptr(x), L.ptr(wp), L.ptr(bias), L.ptr(alpha), L.ptr(residual), L.ptr(out),
                                       L.ptr(ws), ws.numel() * 4 if ws is not None else 0, s)

    rc = _launch("conv_fwd", d, lambda: run(y), bias is not None, residual is not None)
    if rc == 0:
        _attach_range(y, rout)
        _shadow("fwd", d, lambda o: run(o[0], False), [y])
    return rc


def _unit_fwd(d3, d1, x, wp3, wp1, h, y, s):
    """Fused Residual(DilatedUnit) forward (rh_residual_unit_fwd_f32); ``h`` None = inference (never written)."""
    rin = ry = rh = None
    if _RANGES:
        rin, ry = _range_of(x, s, "unit x"), _new_range(x.device)
        rh = _new_range(x.device) if h is not None else None

    def run(o_y, o_h):
        if rin is not None:
            L.lib.rh_x6_set_ranges(None, L.ptr(rin), L.ptr(ry), L.ptr(rh))
        return L.lib.rh_residual_unit_fwd_f32(C.byref(d3), C.byref(d1), L.ptr(x), L.ptr(wp3), L.ptr(wp1), L.ptr(o_h), L.ptr(o_y), s)

    if _PLAN_LOG is not None:
        _PLAN_LOG.append((2, (d3.c_in, d3.c_out, d3.kernel, d3.stride, d3.dilation, d3.l_in, d3.transposed), (3, d3.c_in // 32, 2, 1, 1, 1, 0, 0)))
    if _PROFILE is None:
        rc = run(y, h)
    else:
        f3, b3 = _conv_cost(d3)
        f1, b1 = _conv_cost(d1)
        rc = _timed("conv_fwd[x6]", f3 + f1, b3 + b1, lambda: run(y, h))
    if rc == 0:
        _attach_range(y, ry)
        _attach_range(h, rh)
    if rc == 0 and _SHADOW is not None:
        # the two-launch path on the exact-f32 kernels, same operands
        h2, y2 = torch.empty_like(x), torch.empty_like(x)
        with _ExactF32():
            nf3 = L.lib.rh_conv1d_fwd_workspace_bytes(C.byref(d3))
            nf1 = L.lib.rh_conv1d_fwd_workspace_bytes(C.byref(d1))
            ws = _ws(max(nf3, nf1, 4), x.device)
            L.check(L.lib.rh_conv1d_fwd_f32(C.byref(d3), L.ptr(x), L.ptr(wp3), None, None, None, L.ptr(h2), L.ptr(ws), ws.numel() * 4, s),
                    "shadow unit k3")
            L.check(L.lib.rh_conv1d_fwd_f32(C.byref(d1), L.ptr(h2), L.ptr(wp1), None, None, L.ptr(x), L.ptr(y2), L.ptr(ws), ws.numel() * 4, s),
                    "shadow unit k1")
        key = (d3.c_in, d3.c_out, d3.kernel, d3.stride, d3.dilation, d3.l_in, d3.transposed, d3.batch)
        _SHADOW.append(("unit", key, _rel(y, y2)))
        if h is not None:
            _SHADOW.append(("unit_h", key, _rel(h, h2)))
    return rc


def _dgrad(d, dy, wp, x, alpha, add, dx, s):
    _log_plan(d, 1, False, add is not None)
    ws = _ws(L.lib.rh_conv1d_bwd_data_workspace_bytes(C.byref(d)), dx.device)
    rin = rout = None
    if _ranges_on():
        if L.lib.rh_conv1d_kernel_family(C.byref(d), 1, 0, int(add is not None)) == 1:
            rin = _range_of(dy, s, "dgrad dy")
        rout = _new_range(dx.device)

    def run(out, ranges=True):
        if ranges and rout is not None:
            L.lib.rh_x6_set_ranges(None, L.ptr(rin), L.ptr(rout), None)
        return L.lib.rh_conv1d_bwd_data_f32(C.byref(d), L.ptr(dy), L.ptr(wp), L.ptr(x), L.ptr(alpha), L.ptr(add), L.ptr(out),
                                            L.ptr(ws), ws.numel() * 4 if ws is not None else 0, s)

    rc = _launch("conv_dgrad", d, lambda: run(dx), False, add is not None)
    if rc == 0:
        _attach_range(dx, rout)
        _shadow("dgrad", d, lambda o: run(o[0], False), [dx])
    return rc


# ---- weight-gradient branch on a side stream -----------------------------------------------------------------------------
# In the backward pass the weight-gradient branch of a layer (wgrad kernel -> ordered reduction of its K-slice partials ->
# weight-norm backward: one MFMA kernel with a bandwidth-bound store tail, then two small bandwidth-bound kernels -- the
# last of them collected over all layers of the branch into one launch at the join, _WN_PENDING below) only
# feeds the optimizer, while the data-gradient chain is what the next layer waits for.  The branch (RH_BWD_SIDE_STREAM=0
# disables) is enqueued on a second HIP stream: it forks after the kernels that produce its operands and is joined back (a) at the
# end of the backward pass (autograd final callback) and (b) before a data-parallel bucket leaves (rave_amd.ddp).  Recorded
# into the step's hipGraph the two streams become parallel branches of the graph.
_SIDE = {}
_SIDE_PENDING = [None]
_SIDE_HOLD = []        # incoming gradient tensors the branch still reads (see _OnSide): released when the branch is joined


def _side_enabled() -> bool:
    import os
    return os.environ.get("RH_BWD_SIDE_STREAM", "1") != "0"


def _side_stream(device):
    """The stream of the weight-gradient branch of the backward pass (one per device)."""
    k = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE.get(k)
    if st is None:
        st = _SIDE[k] = torch.cuda.Stream(device=device)
    return st


# Weight-norm backward of the layers whose weight gradient ran on the side stream: collected here and run as ONE launch
# (rh_weight_norm_bwd_batched_f32) when the branch is joined -- at the end of the backward pass or when a data-parallel
# bucket leaves -- instead of one 4-6 us latency-bound launch per layer (56 per v2 step).  The dv / dg tensors handed to
# autograd are filled by that launch: like every gradient of the side stream they are valid once the branch is joined.
# RH_WN_BATCH=0: one launch per layer, at once.
_WN_PENDING = []
import os as _os
_WN_BATCH_MAX = int(_os.environ.get("RH_WN_BATCH_MAX", "64"))      # layers per launch (= the table size of the kernel): a pass with more pending layers launches early, on the side stream.
#                         (16 / 8 per launch -- so that only the last, small layers are left for the join -- measured the same
#                         step time as one launch at the join: 10.18-10.24 vs 10.16-10.20 ms)


def _wn_batch_enabled() -> bool:
    import os
    return os.environ.get("RH_WN_BATCH", "1") != "0"


# The ordered reductions of the weight gradients' K-slice partials, collected the same way (round 6: 56 launches of 9 - 15 us per
# v2 step, one behind every split weight-gradient kernel of the side stream): a weight-gradient call of the side stream leaves its
# partials in a scratch of its own (rh_defer_reduce) and every RH_REDUCE_BATCH layers -- and whenever the gradients are needed: the
# collected weight-norm launch, a data-parallel bucket, the join -- ONE launch reduces them all, each in the order its own launch
# would have used (bit-identical; RH_REDUCE_BATCH=0: one launch per layer, at once).
_RED_PENDING = []       # (ReduceItem, scratch tensor, dw tensor): the tensors are kept alive until the batch has been launched
_RED_BATCH = 8


def _reduce_batch() -> int:
    """Layers per batched reduction; 0 = one launch per layer.  Default: 8 while a hipGraph is being recorded, 0 in eager steps --
    there the per-layer scratch allocation and the extra calls cost more host time than the launches they save (measured: the
    eager v2 step 9.5 -> 11.1 ms with batching, the replayed step unchanged at 232 instead of 281 dispatches).
    RH_REDUCE_BATCH=n forces n in both modes (read per call: tests)."""
    import os
    e = os.environ.get("RH_REDUCE_BATCH")
    if e is not None:
        return int(e)
    return _RED_BATCH if torch.cuda.is_current_stream_capturing() else 0


def _flush_reduce_pending(stream_ptr) -> None:
    items, _RED_PENDING[:] = list(_RED_PENDING), []
    if not items:
        return
    arr = (L.ReduceItem * len(items))()
    for a, (it, _, _) in zip(arr, items):
        a.part, a.out, a.n, a.Z = it.part, it.out, it.n, it.Z
    L.check(L.lib.rh_reduce_partials_batched_f32(arr, len(items), stream_ptr), "reduce_partials_batched")


def _flush_wn_pending(stream_ptr) -> None:
    _flush_reduce_pending(stream_ptr)        # (the weight-norm backward reads the reduced gradients)
    items, _WN_PENDING[:] = list(_WN_PENDING), []
    if not items:
        return
    arr = (L.WnBwdItem * len(items))()
    for a, (dw, v, g, norms, dv, dg) in zip(arr, items):
        a.dw, a.v, a.g, a.norms, a.dv, a.dg = L.ptr(dw), L.ptr(v), L.ptr(g), L.ptr(norms), L.ptr(dv), L.ptr(dg)
        a.rows = v.shape[0]
        a.cols = v.numel() // max(v.shape[0], 1)
    L.check(L.lib.rh_weight_norm_bwd_batched_f32(arr, len(items), stream_ptr), "weight_norm_bwd_batched")


def side_stream_for_collective(device):
    """Data-parallel bucket leaving while the backward pass still runs (rave_amd.ddp.GradReducer._launch): returns the stream
    the bucket's all-reduce must be ISSUED FROM so that it is ordered behind every gradient written so far -- the
    weight-gradient side stream, after the collected weight-norm launch of the branch has been enqueued there and after it
    has been made to wait for the calling (compute) stream -- or None when the branch has nothing in flight.  The compute
    stream itself is NOT made to wait: the data-gradient chain keeps running while the branch finishes and the collective
    starts (round 4 joined the branch into the compute stream at every bucket boundary)."""
    pend = _SIDE_PENDING[0]
    if pend is None:
        return None
    side = pend[1]
    if _WN_PENDING or _RED_PENDING:
        with torch.cuda.stream(side):
            _flush_wn_pending(L.stream())
    side.wait_stream(torch.cuda.current_stream(device))
    return side


def _graph_task_id() -> int:
    """Id of the autograd graph task (backward pass) this thread is executing, -1 outside one."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    return int(f()) if f is not None else -1


def join_side_streams() -> None:
    """The calling stream waits for everything enqueued on the weight-gradient side stream (after the collected
    weight-norm backward launches of that branch have been enqueued there)."""
    pend = _SIDE_PENDING[0]
    if pend is not None:
        main, side = pend[0], pend[1]
        if _WN_PENDING or _RED_PENDING:
            with torch.cuda.stream(side):
                _flush_wn_pending(L.stream())
        main.wait_stream(side)
        torch.cuda.current_stream().wait_stream(side)
        _SIDE_PENDING[0] = None
        _SIDE_HOLD.clear()


def abandon_side_streams(rejoin: bool = False) -> None:
    """Forget the side-stream bookkeeping of a pass that will never be joined (a step whose graph capture was refused midway):
    no kernel is launched -- the work recorded so far dies with the capture.  rejoin: the calling stream waits for the side
    streams first, so that a stream capture that forked onto them can be ended."""
    if rejoin:
        cur = torch.cuda.current_stream()
        for st in _SIDE.values():
            try:
                cur.wait_stream(st)
            except Exception:             # noqa: BLE001 -- a side stream that never joined the capture
                pass
    _SIDE_PENDING[0] = None
    _SIDE_HOLD.clear()
    _WN_PENDING.clear()
    _RED_PENDING.clear()


def _note_use(ctx, *params) -> None:
    """Forward side of the "exactly one pending use" rule of the side stream: a parameter that enters two nodes of one
    graph gets its two gradients ADDED by autograd on the compute stream -- which must not happen to a tensor the side
    stream is still writing.  Every differentiable use bumps a counter on the parameter, every backward takes it down."""
    ps = [p for p in params if p is not None and p.requires_grad]
    for p in ps:
        p._rh_pending = getattr(p, "_rh_pending", 0) + 1
        if p._rh_pending > 1:
            p._rh_shared = True          # stays set until every pending use has been taken down
    ctx.rh_params = ps


def _single_use(ctx) -> bool:
    """Backward side: True iff every parameter of this node has exactly this one pending use and no gradient yet (then the
    gradient returned here is adopted by autograd without being read).  Always takes the counters down."""
    ok = True
    for p in getattr(ctx, "rh_params", ()):
        n = getattr(p, "_rh_pending", 1)
        ok = ok and n == 1 and not getattr(p, "_rh_shared", False)
        # an EXISTING gradient (accumulation over several backward() calls, zero_grad(set_to_none=False), the other
        # optimizer's parameters in a GAN step) is accumulated into by AccumulateGrad on the COMPUTE stream -- it must not
        # read a tensor the side stream is still writing; likewise the data-parallel hook copies a gradient
        # into its bucket view on the compute stream when the view was not the one written (slot active but not fresh)
        if p.grad is not None:
            ok = False
        slot = getattr(p, "_rh_grad_slot", None)
        if slot is not None and len(slot) > 2 and slot[2] and not slot[1]:
            ok = False
        p._rh_pending = max(n - 1, 0)
        if p._rh_pending == 0:
            p._rh_shared = False
    return ok


class _OnSide:
    """``with _OnSide(device, tensors...)``: the body is enqueued on the side stream after everything already enqueued on
    the current stream; ``tensors`` are kept alive for it (caching-allocator stream bookkeeping)."""

    def __init__(self, device, *tensors, allow: bool = True, hold=()):
        """``hold``: gradient tensors RECEIVED from autograd that the body reads.  The same tensor object may sit in another
        node's input buffer (``y = conv(x) + other``: AddBackward hands ONE tensor to both branches), and autograd
        accumulates into a buffered gradient IN PLACE once it is the only holder -- on the compute stream, while the side
        stream still reads it.  A reference kept until the branch is joined makes that accumulation take the
        out-of-place form."""
        self.device = device
        self.tensors = [t for t in tensors if t is not None]
        self.hold = [t for t in hold if t is not None]
        import os
        self.active = allow and _side_enabled()

    def __enter__(self):
        if not self.active:
            return self
        self.main = torch.cuda.current_stream(self.device)
        self.side = _side_stream(self.device)
        self.side.wait_stream(self.main)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def keep(self, *tensors):
        self.tensors += [t for t in tensors if t is not None]

    def __exit__(self, *exc):
        if not self.active:
            return False
        self.ctx.__exit__(*exc)
        for t in self.tensors:
            t.record_stream(self.side)
        # The join is queued once per BACKWARD PASS (graph task).  A pass that raised after queueing never ran its callback
        # and left the pending state behind: a pending entry of ANOTHER pass is joined here and now, so that its
        # collected weight-norm launch still runs and a later pass queues its own join.
        task = _graph_task_id()
        pend = _SIDE_PENDING[0]
        if pend is not None and pend[2] != task:
            join_side_streams()
            pend = None
        # (after the stale join, which clears _SIDE_HOLD: this node's holds must survive until ITS join -- ADVICE r5)
        if os.environ.get("RH_SIDE_HOLD", "1") != "0":      # (0: the unprotected form, for the test that shows the race)
            _SIDE_HOLD.extend(self.hold)
        if pend is None:
            _SIDE_PENDING[0] = (self.main, self.side, task)
            try:        # end of this backward pass: the compute stream waits for the branch
                torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
            except RuntimeError:
                join_side_streams()       # not inside a backward pass (direct call): join at once
        return False


def _arm_wgrad_ranges(dy, x, s, d=None) -> None:
    """Range slots of both operands for the next weight-gradient call (the f16 weight-gradient kernel converts both)."""
    if _RANGES and os.environ.get("RH_WGRAD_X6", "1") != "0" and (d is None or L.lib.rh_conv1d_bwd_weight_kernel_family(C.byref(d)) == 1):
        L.lib.rh_x6_set_ranges(L.ptr(_range_of(dy, s, "wgrad dy")), L.ptr(_range_of(x, s, "wgrad x")), None, None)


def _wgrad(d, dy, x, alpha, dw, db, ws, nbytes, s, defer=None):
    """``defer``: a ReduceItem -- the call may leave the reduction of its K-slice partials to the caller (rh_defer_reduce)."""
    def run(o_dw, o_db):
        _arm_wgrad_ranges(dy, x, s, d)
        if defer is not None:
            L.lib.rh_defer_reduce(C.byref(defer))
        return L.lib.rh_conv1d_bwd_weight_f32(C.byref(d), L.ptr(dy), L.ptr(x), L.ptr(alpha), L.ptr(o_dw), L.ptr(o_db), L.ptr(ws),
                                              nbytes, s)

    rc = _launch("conv_wgrad", d, lambda: run(dw, db))
    if rc == 0 and _SHADOW is not None:
        def run_exact(o):
            L.lib.rh_x6_set_ranges(None, None, None, None)
            # the exact-f32 kernels plan their own K slices: their scratch is sized under THEIR plan (it used to fit into the
            # bf16x6 path's by accident, while that path cut K into twice as many slices)
            nb2 = L.lib.rh_conv1d_workspace_bytes(C.byref(d))
            ws2 = torch.empty(max(nb2, 4) // 4, device=dy.device, dtype=torch.float32)
            return L.lib.rh_conv1d_bwd_weight_f32(C.byref(d), L.ptr(dy), L.ptr(x), L.ptr(alpha), L.ptr(o[0]), L.ptr(o[1]), L.ptr(ws2),
                                                  nb2, s)
        _shadow("wgrad", d, run_exact, [dw, db])
    return rc


def _shapes(x: Tensor, weight: Tensor, g: ConvGeom):
    """Returns (batch, c_in, c_out, l_in, l_out, k, in_valid, out_shape)."""
    k = weight.shape[2]
    if g.transposed:
        c_in, c_out = weight.shape[0], weight.shape[1]
    else:
        c_out, c_in = weight.shape[0], weight.shape[1]
    b = x.shape[0]
    if x.shape[1] != c_in:
        raise RuntimeError(f"rave_amd conv: input has {x.shape[1]} channels, weight expects {c_in}")
    in_valid = 0
    if g.inner > 1:
        if g.fold:  # (B, C, T) waveform, zero-padded to a multiple of `inner` in-kernel
            t = x.shape[2]
            l_in = -(-t // g.inner)
            in_valid = t if l_in * g.inner != t else 0
        else:       # (B, C, H, W)
            if x.dim() != 4 or x.shape[3] != g.inner:
                raise RuntimeError("rave_amd conv: expected (B, C, H, W) with W == inner")
            l_in = x.shape[2]
        l_out = g.out_len(l_in, k)
        out_shape = (b, c_out, l_out, g.inner)
    else:
        l_in = x.shape[2]
        l_out = g.out_len(l_in, k)
        out_shape = (b, c_out, l_out)
    return b, c_in, c_out, l_in, l_out, k, in_valid, out_shape


def _pack(d, w3, g, need_bwd, dev, s, prepacked=None):
    """Packed weight copies for one conv; with ``g`` the weight norm g*v/||v|| is folded into the
    repack (rh_conv1d_pack_wn_f32).  Returns (wp_fwd, wp_bwd, norms).  ``prepacked`` = the same
    triple already produced for the current weights by rave_amd.prep.WeightPrep (two launches for
    the whole model instead of two per layer)."""
    if prepacked is not None:
        return prepacked
    dref = C.byref(d)
    wp_f = torch.empty(L.lib.rh_conv1d_packed_floats(dref, 0), device=dev, dtype=torch.float32)
    wp_b = torch.empty(L.lib.rh_conv1d_packed_floats(dref, 1), device=dev, dtype=torch.float32) if need_bwd else None
    norms = None
    if g is not None:
        rows = w3.shape[0]
        ns = torch.empty(2, rows, device=dev, dtype=torch.float32)
        norms = ns[0]
        L.check(L.lib.rh_conv1d_pack_wn_f32(dref, L.ptr(w3), L.ptr(g), L.ptr(ns[0]), L.ptr(ns[1]), L.ptr(wp_f),
                                            L.ptr(wp_b), s), "conv1d_pack_wn")
    else:
        L.check(L.lib.rh_conv1d_pack_f32(dref, L.ptr(w3), L.ptr(wp_f), L.ptr(wp_b), s), "conv1d_pack")
    return wp_f, wp_b, norms


def _slot_of(p):
    """The parameter's gradient slot ([view into a flat all-reduce bucket, fresh flag]) installed by
    rave_amd.ddp.GradReducer, or None."""
    return getattr(p, "_rh_grad_slot", None) if p is not None else None


def _grad_out(slot, shape, device) -> Tensor:
    """Where a parameter gradient is written: the parameter's bucket view while the data-parallel reducer has marked it
    fresh for this step (autograd then adopts the view as ``p.grad``: no pack / copy-back around the all-reduce), else a
    new tensor.  The first writer of a step takes the view; a parameter used twice accumulates through autograd."""
    if slot is not None and slot[1] and tuple(slot[0].shape) == tuple(shape) and slot[0].device == device:
        slot[1] = False
        # a fresh alias of the view: autograd adopts an incoming gradient (instead of cloning it) only if nothing else
        # holds the tensor object; the reducer keeps the view itself
        return slot[0].detach()
    return torch.empty(tuple(shape), device=device, dtype=torch.float32)


def _wn_bwd(dw, v, g, norms, s, slot_v=None, slot_g=None):
    rows = v.shape[0]
    cols = v.numel() // max(rows, 1)
    dv = _grad_out(slot_v, v.shape, v.device)
    dg = _grad_out(slot_g, g.shape, g.device)
    L.check(L.lib.rh_weight_norm_bwd_f32(L.ptr(dw), L.ptr(v), L.ptr(g), L.ptr(norms), rows, cols, L.ptr(dv), L.ptr(dg), s),
            "weight_norm_bwd")
    return dv, dg


def _wgrad_wn(d, dy, x, alpha, dw, db, v, g, norms, ws, nbytes, s, slot_v=None, slot_g=None, side=None):
    """Weight gradient of a weight-normed conv: (dv, dg).  On the side stream (``side.active``) only the weight gradient is
    launched here; weight norm's backward joins the batch that runs when the branch is joined (_WN_PENDING).  Otherwise ONE C
    call (rh_conv1d_bwd_weight_wn_f32: weight gradient, ordered reduction, weight-norm backward).  The instrumented modes
    (per-launch profile, exact-f32 shadow run) keep the two separate calls they account for."""
    if _PROFILE is not None or _SHADOW is not None:
        L.check(_wgrad(d, dy, x, alpha, dw, db, ws, nbytes, s), "conv1d_bwd_weight")
        return _wn_bwd(dw, v, g, norms, s, slot_v, slot_g)
    dv = _grad_out(slot_v, v.shape, v.device)
    dg = _grad_out(slot_g, g.shape, g.device)
    if side is not None and side.active and _wn_batch_enabled():
        nred = _reduce_batch()
        if nred > 0:
            # a scratch of its own (the caller's may be shared with the next weight gradient): the partials stay until the batch runs
            own = torch.empty(max(nbytes, 4) // 4, device=dy.device, dtype=torch.float32)
            item = L.ReduceItem()
            L.check(_wgrad(d, dy, x, alpha, dw, db, own, nbytes, s, defer=item), "conv1d_bwd_weight")
            if item.Z > 1:
                _RED_PENDING.append((item, own, dw))
                if len(_RED_PENDING) >= nred:
                    _flush_reduce_pending(s)
        else:
            L.check(_wgrad(d, dy, x, alpha, dw, db, ws, nbytes, s), "conv1d_bwd_weight")
        # (aliases of dv / dg keep the storage alive until the batch has run; the tensor OBJECTS handed to autograd must have
        # no other holder, or AccumulateGrad clones -- i.e. reads -- them instead of adopting them)
        _WN_PENDING.append((dw, v, g, norms, dv.detach(), dg.detach()))
        if len(_WN_PENDING) >= _WN_BATCH_MAX:      # (we are on the side stream here: the batch runs beside the data-gradient chain)
            _flush_wn_pending(s)
        return dv, dg
    _arm_wgrad_ranges(dy, x, s, d)
    L.check(L.lib.rh_conv1d_bwd_weight_wn_f32(C.byref(d), L.ptr(dy), L.ptr(x), L.ptr(alpha), L.ptr(v), L.ptr(g), L.ptr(norms),
                                              L.ptr(dw), L.ptr(dv), L.ptr(dg), L.ptr(db), L.ptr(ws), nbytes, s),
            "conv1d_bwd_weight_wn")
    return dv, dg


class _ConvFn(torch.autograd.Function):
    """y = conv(act(x), w) + bias + residual, w = weight (g is None) or g*weight/||weight|| (weight norm
    folded in).  rh_conv1d_fwd_f32 and its gradients."""

    @staticmethod
    def forward(ctx, x, weight, g, bias, alpha, residual, geom: ConvGeom, prepacked=None):
        ctx.slots = (_slot_of(weight), _slot_of(g), _slot_of(bias))
        _note_use(ctx, weight, g, bias)
        _log_gate((weight, g, bias), x, geom.act)
        x = _chk(x, "x"); weight = _chk(weight, "weight"); g = _chk(g, "weight_g"); bias = _chk(bias, "bias")
        alpha = _chk(alpha, "alpha"); residual = _chk(residual, "residual")
        w3 = weight.reshape(weight.shape[0], weight.shape[1], -1) if weight.dim() == 4 else weight
        b, c_in, c_out, l_in, l_out, k, in_valid, out_shape = _shapes(x, w3, geom)
        d = _desc(geom, b, c_in, c_out, l_in, l_out, k, in_valid)
        s = L.stream()
        wp_f, wp_b, norms = _pack(d, w3, g, ctx.needs_input_grad[0], x.device, s, prepacked)
        y = torch.empty(out_shape, device=x.device, dtype=torch.float32)
        if residual is not None and residual.shape != y.shape:
            raise RuntimeError(f"rave_amd conv: residual shape {tuple(residual.shape)} != output {tuple(y.shape)}")
        L.check(_fwd(d, x, wp_f, bias, alpha, residual, y, s), "conv1d_fwd")
        ctx.save_for_backward(x, wp_b, alpha, weight if g is not None else None, g, norms,
                              y if geom.out_act != ACT_NONE else None)
        ctx.x_range = getattr(x, "_rh_range", None)
        ctx.d = d
        ctx.wshape = tuple(weight.shape)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp_b, alpha, v, g, norms, y_act = ctx.saved_tensors
        if ctx.x_range is not None and getattr(x, "_rh_range", None) is None:
            x._rh_range = ctx.x_range          # (the saved tensor may come back as a new object)
        d = ctx.d
        dref = C.byref(d)
        dy = _chk(dy, "dy")
        s = L.stream()
        dx = dw = dg = db = dres = None
        # side stream only if (a) nothing autograd may touch on the compute stream is still in use over there: dy is handed
        # on as the residual's gradient below, and autograd accumulates INTO such tensors in place; (b) the parameters'
        # gradients are adopted unread (one pending use each)
        side_ok = _single_use(ctx) and not (ctx.has_res and ctx.needs_input_grad[5])
        dy_in = dy if y_act is None else None      # (with an output activation the branch reads gpre, a tensor of this node)
        if y_act is not None:      # output activation: every gradient below is taken w.r.t. the pre-activation
            gpre = torch.empty_like(dy)
            L.check(L.lib.rh_act_bwd_f32(L.ptr(dy), L.ptr(y_act), d.out_act, d.out_slope, dy.numel(), L.ptr(gpre), s),
                    "act_bwd")
            dy = gpre
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.check(_dgrad(d, dy, wp_b, x, alpha, None, dx, s), "conv1d_bwd_data")
        need_w = ctx.needs_input_grad[1] or (g is not None and ctx.needs_input_grad[2])
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        if need_w or need_b:
            slot_w, slot_g, slot_b = ctx.slots
            # without weight norm the weight gradient IS the parameter gradient: written into its bucket view
            dw = _grad_out(slot_w if g is None else None, ctx.wshape, dy.device)
            if need_b:
                db = _grad_out(slot_b, (d.c_out,), dy.device)
            nbytes = L.lib.rh_conv1d_workspace_bytes(dref)
            ws = torch.empty(max(nbytes, 4) // 4, device=dy.device, dtype=torch.float32)
            with _OnSide(dy.device, dy, x, ws, dw, db, v, g, norms, alpha, allow=side_ok, hold=(dy_in,)) as side:
                s2 = L.stream()
                if g is None:
                    L.check(_wgrad(d, dy, x, alpha, dw, db, ws, nbytes, s2), "conv1d_bwd_weight")
                else:
                    dw_full = dw
                    dw, dg = _wgrad_wn(d, dy, x, alpha, dw, db, v, g, norms, ws, nbytes, s2, slot_w, slot_g, side)
                    side.keep(dw_full, dw, dg)
        if alpha is not None and ctx.needs_input_grad[4]:
            raise NotImplementedError("rave_amd: gradient w.r.t. a fused Snake alpha; use rave_amd.blocks.Snake + conv")
        if ctx.has_res and ctx.needs_input_grad[5]:
            dres = dy
        return dx, dw, dg, db, None, dres, None, None


def conv1d(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, *, geom: ConvGeom,
           alpha: Optional[Tensor] = None, residual: Optional[Tensor] = None,
           weight_g: Optional[Tensor] = None, prepacked=None) -> Tensor:
    """``weight_g`` given: ``weight`` is the weight-norm direction v and w = g*v/||v|| (dim 0)."""
    return _ConvFn.apply(x, weight, weight_g, bias, alpha, residual, geom, prepacked)


class _WeightNormFn(torch.autograd.Function):
    """w = g * v / ||v|| over dims != 0 (rh_weight_norm_{fwd,bwd}_f32)."""

    @staticmethod
    def forward(ctx, v, g):
        v = _chk(v, "weight_v"); g = _chk(g, "weight_g")
        rows = v.shape[0]
        cols = v.numel() // max(rows, 1)
        w = torch.empty_like(v)
        norms = torch.empty(rows, device=v.device, dtype=torch.float32)
        L.check(L.lib.rh_weight_norm_fwd_f32(L.ptr(v), L.ptr(g), rows, cols, L.ptr(w), L.ptr(norms), L.stream()),
                "weight_norm_fwd")
        ctx.save_for_backward(v, g, norms)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, norms = ctx.saved_tensors
        dw = _chk(dw, "dw")
        rows = v.shape[0]
        cols = v.numel() // max(rows, 1)
        dv = torch.empty_like(v)
        dg = torch.empty_like(g)
        L.check(L.lib.rh_weight_norm_bwd_f32(L.ptr(dw), L.ptr(v), L.ptr(g), L.ptr(norms), rows, cols,
                                             L.ptr(dv), L.ptr(dg), L.stream()), "weight_norm_bwd")
        return dv, dg


def weight_norm(v: Tensor, g: Tensor) -> Tensor:
    return _WeightNormFn.apply(v, g)


class _ResidualUnitFn(torch.autograd.Function):
    """Residual(DilatedUnit) as ONE autograd node (rave/blocks.py:31-45, 83-112):
        h = conv_k3_dil(act(x), w3);  y = conv_k1(act(h), w1) + x
    with w = g*v/||v|| folded into the weight repack when g3/g1 are given.  Saves only x and h;
    activations, padding, weight norm and the residual add live inside the kernels."""

    @staticmethod
    def forward(ctx, x, w3, g3w, w1, g1w, alpha0, alpha2, g3: ConvGeom, g1: ConvGeom, pre3=None, pre1=None):
        ctx.slots = (_slot_of(w3), _slot_of(g3w), _slot_of(w1), _slot_of(g1w))
        _note_use(ctx, w3, g3w, w1, g1w)
        gate_ids = ((w3, g3w), (w1, g1w))
        x = _chk(x, "x"); w3 = _chk(w3, "w3"); w1 = _chk(w1, "w1"); g3w = _chk(g3w, "g3"); g1w = _chk(g1w, "g1")
        alpha0 = _chk(alpha0, "alpha0"); alpha2 = _chk(alpha2, "alpha2")
        b, c, l = x.shape
        k = w3.shape[2]
        d3 = _desc(g3, b, c, c, l, l, k)
        d1 = _desc(g1, b, c, c, l, l, 1)
        s = L.stream()
        wp3f, wp3b, n3 = _pack(d3, w3, g3w, True, x.device, s, pre3)
        wp1f, wp1b, n1 = _pack(d1, w1, g1w, True, x.device, s, pre1)
        y = torch.empty_like(x)
        if alpha0 is None and alpha2 is None and L.lib.rh_residual_unit_fused(C.byref(d3), C.byref(d1)) == 1:
            # one launch (unit_x6.hip): h stays in registers; it is only written when a backward pass will need it
            h = torch.empty_like(x) if any(ctx.needs_input_grad[:5]) else None
            L.check(_unit_fwd(d3, d1, x, wp3f, wp1f, h, y, s), "residual_unit_fwd")
        else:
            h = torch.empty_like(x)
            L.check(_fwd(d3, x, wp3f, None, alpha0, None, h, s), "unit k3")
            L.check(_fwd(d1, h, wp1f, None, alpha2, x, y, s), "unit k1")
        _log_gate(gate_ids[0], x, g3.act)
        if h is not None:
            _log_gate(gate_ids[1], h, g1.act)
        ctx.save_for_backward(x, h, wp3b, wp1b, alpha0, alpha2, w3 if g3w is not None else None, g3w, n3,
                              w1 if g1w is not None else None, g1w, n1)
        ctx.d3, ctx.d1 = d3, d1
        ctx.w3shape, ctx.w1shape = tuple(w3.shape), tuple(w1.shape)
        ctx.x_range = getattr(x, "_rh_range", None)
        ctx.h_range = getattr(h, "_rh_range", None) if h is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, wp3b, wp1b, alpha0, alpha2, v3, g3w, n3, v1, g1w, n1 = ctx.saved_tensors
        if ctx.x_range is not None and getattr(x, "_rh_range", None) is None:
            x._rh_range = ctx.x_range
        if ctx.h_range is not None and getattr(h, "_rh_range", None) is None:
            h._rh_range = ctx.h_range
        d3, d1 = ctx.d3, ctx.d1
        r3, r1 = C.byref(d3), C.byref(d1)
        dy = _chk(dy, "dy")
        s = L.stream()
        dev = dy.device
        if (alpha0 is not None and ctx.needs_input_grad[5]) or (alpha2 is not None and ctx.needs_input_grad[6]):
            raise NotImplementedError("rave_amd: Snake alpha gradient in the fused residual unit")
        dh = torch.empty_like(h)
        L.check(_dgrad(d1, dy, wp1b, h, alpha2, None, dh, s), "unit k1 dgrad")
        dw1 = dw3 = dg1 = dg3 = dx = None
        nb1 = L.lib.rh_conv1d_workspace_bytes(r1)
        nb3 = L.lib.rh_conv1d_workspace_bytes(r3)
        ws = torch.empty(max(nb1, nb3, 4) // 4, device=dev)
        # both weight-gradient branches (operands: dy, h, dh, x -- all produced by now) may run beside the k3 data gradient
        # (dy is only read here -- the residual gradient is added inside the k3 data-gradient kernel -- and dx is a new tensor)
        with _OnSide(dev, dy, h, dh, x, ws, v3, g3w, n3, v1, g1w, n1, alpha0, alpha2, allow=_single_use(ctx), hold=(dy,)) as side:
            s2 = L.stream()
            if ctx.needs_input_grad[3] or (g1w is not None and ctx.needs_input_grad[4]):
                s3w, s3g, s1w, s1g = ctx.slots
                dw1 = _grad_out(s1w if g1w is None else None, ctx.w1shape, dev)
                side.keep(dw1)
                if g1w is None:
                    L.check(_wgrad(d1, dy, h, alpha2, dw1, None, ws, nb1, s2), "unit k1 wgrad")
                else:
                    dw1, dg1 = _wgrad_wn(d1, dy, h, alpha2, dw1, None, v1, g1w, n1, ws, nb1, s2, s1w, s1g, side)
                    side.keep(dw1, dg1)
            if ctx.needs_input_grad[1] or (g3w is not None and ctx.needs_input_grad[2]):
                s3w, s3g, s1w, s1g = ctx.slots
                dw3 = _grad_out(s3w if g3w is None else None, ctx.w3shape, dev)
                side.keep(dw3)
                if g3w is None:
                    L.check(_wgrad(d3, dh, x, alpha0, dw3, None, ws, nb3, s2), "unit k3 wgrad")
                else:
                    dw3, dg3 = _wgrad_wn(d3, dh, x, alpha0, dw3, None, v3, g3w, n3, ws, nb3, s2, s3w, s3g, side)
                    side.keep(dw3, dg3)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            # dx = act'(x) * dgrad_k3(dh) + dy   (residual gradient fused as `add`)
            L.check(_dgrad(d3, dh, wp3b, x, alpha0, dy, dx, s), "unit k3 dgrad")
        return dx, dw3, dg3, dw1, dg1, None, None, None, None, None, None


def residual_unit(x, w3, w1, g3: ConvGeom, g1: ConvGeom, alpha0=None, alpha2=None, w3_g=None, w1_g=None,
                  pre3=None, pre1=None) -> Tensor:
    """``w3_g`` / ``w1_g`` given: w3 / w1 are weight-norm directions (v) and the gains are folded in."""
    return _ResidualUnitFn.apply(x, w3, w3_g, w1, w1_g, alpha0, alpha2, g3, g1, pre3, pre1)


# --------------------------------------------------------------------------- general Conv2d
def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _conv2d_cost(d: "L.Conv2dDesc"):
    flops = 2.0 * d.batch * d.c_out * d.c_in * d.kh * d.kw * d.h_out * d.w_out
    elems = d.batch * (d.c_in * d.h_in * d.w_in + d.c_out * d.h_out * d.w_out) + d.c_out * d.c_in * d.kh * d.kw
    return flops, 4.0 * elems


def _launch2(kind: str, d, fn):
    if _PROFILE is None:
        return fn()
    if kind in ("conv2d_fwd", "conv2d_dgrad"):     # which kernel family does this launch take? (rh_conv2d_plan_info)
        info = (C.c_int64 * 16)()
        if L.lib.rh_conv2d_plan_info(C.byref(d), 0 if kind == "conv2d_fwd" else 1, info) == 0:
            kind += _FAMILY.get(int(info[0]), "[f32]")
    elif kind == "conv2d_wgrad":
        kind += _FAMILY.get(L.lib.rh_conv2d_bwd_weight_kernel_family(C.byref(d)), "[f32]")
    f, b = _conv2d_cost(d)
    return _timed(kind, f, b, fn)


def _conv2d_ranges(d, which: int, t: Tensor, dev, s, tag: str):
    """(input slot, output slot) of a 2-D forward (which = 0) / data-gradient (1) launch in the f16 build: the input's only where
    the matrix-core kernel (conv2d_x6.hip, family 1) will read it, the output's always (every kernel family fills it)."""
    if not _ranges_on():
        return None, None
    info = (C.c_int64 * 16)()
    fam = int(info[0]) if L.lib.rh_conv2d_plan_info(C.byref(d), which, info) == 0 else 0
    rin = _range_of(t, s, tag) if fam == 1 else None
    return rin, _new_range(dev)


class _Conv2dFn(torch.autograd.Function):
    """y = act(conv2d(x, w) + bias) with zero padding (rh_conv2d_*_f32): torch.nn.Conv2d followed by the
    LeakyReLU the reference's discriminators put after it (rave/discriminator.py:23-51,
    rave/descript_discriminator.py:22-27)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, act: int, slope: float):
        x = _chk(x, "x"); weight = _chk(weight, "weight"); bias = _chk(bias, "bias")
        if x.dim() != 4 or weight.dim() != 4:
            raise RuntimeError("rave_amd conv2d: expects x (B,C,H,W) and weight (Co,Ci,kh,kw)")
        b, c_in, h_in, w_in = x.shape
        c_out, c_in_w, kh, kw = weight.shape
        if c_in != c_in_w:
            raise RuntimeError(f"rave_amd conv2d: input has {c_in} channels, weight expects {c_in_w}")
        (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
        h_out = (h_in + 2 * ph - dh * (kh - 1) - 1) // sh + 1
        w_out = (w_in + 2 * pw - dw * (kw - 1) - 1) // sw + 1
        d = L.Conv2dDesc(batch=b, c_in=c_in, c_out=c_out, h_in=h_in, w_in=w_in, h_out=h_out, w_out=w_out, kh=kh, kw=kw,
                         sh=sh, sw=sw, dh=dh, dw=dw, ph=ph, pw=pw, act=act, act_slope=slope)
        dref = C.byref(d)
        s = L.stream()
        dev = x.device
        nf = L.lib.rh_conv2d_packed_floats(dref, 0)
        if nf < 0:
            L.check(-1, "conv2d geometry")
        wp_f = torch.empty(nf, device=dev, dtype=torch.float32)
        wp_b = torch.empty(L.lib.rh_conv2d_packed_floats(dref, 1), device=dev, dtype=torch.float32) \
            if ctx.needs_input_grad[0] else None
        L.check(L.lib.rh_conv2d_pack_f32(dref, L.ptr(weight), L.ptr(wp_f), L.ptr(wp_b), s), "conv2d_pack")
        y = torch.empty(b, c_out, h_out, w_out, device=dev, dtype=torch.float32)
        rin, rout = _conv2d_ranges(d, 0, x, dev, s, "conv2d x")

        def run_fwd():
            if rout is not None:
                L.lib.rh_x6_set_ranges(None, L.ptr(rin), L.ptr(rout), None)
            return L.lib.rh_conv2d_fwd_f32(dref, L.ptr(x), L.ptr(wp_f), L.ptr(bias), L.ptr(y), s)

        L.check(_launch2("conv2d_fwd", d, run_fwd), "conv2d_fwd")
        _attach_range(y, rout)
        ctx.save_for_backward(x, y if act != ACT_NONE else None, wp_b)
        ctx.d = d
        ctx.wshape = tuple(weight.shape)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, wp_b = ctx.saved_tensors
        d = ctx.d
        dy = _chk(dy, "dy")
        s = L.stream()
        dx = dw = db = None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if y is not None:
            # one elementwise pass forms dy * act'(y); the gradient kernels then run activation-free.  With a bias the SAME
            # pass also reduces the bias gradient (rh_act_bwd_bias_f32: one sweep instead of act_bwd + bias_grad)
            g = torch.empty_like(dy)
            if need_b:
                db = torch.empty(d.c_out, device=dy.device, dtype=torch.float32)
                nb = L.lib.rh_act_bwd_bias_workspace_bytes(d.c_out)
                wsb = torch.empty(max(nb, 4) // 4, device=dy.device, dtype=torch.float32)
                rg = _new_range(dy.device) if _ranges_on() else None     # the same sweep leaves max |g| for the f16 kernels
                if rg is not None:
                    L.lib.rh_x6_set_ranges(None, None, L.ptr(rg), None)
                L.check(L.lib.rh_act_bwd_bias_f32(L.ptr(dy), L.ptr(y), d.act, d.act_slope, d.batch, d.c_out, d.h_out * d.w_out,
                                                  L.ptr(g), L.ptr(db), L.ptr(wsb), nb, s), "act_bwd_bias")
                _attach_range(g, rg)
                need_b = False
            else:
                L.check(L.lib.rh_act_bwd_f32(L.ptr(dy), L.ptr(y), d.act, d.act_slope, dy.numel(), L.ptr(g), s), "act_bwd")
            dy, y = g, None
            d = L.Conv2dDesc.from_buffer_copy(d)
            d.act = ACT_NONE
        dref = C.byref(d)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            rin, rout = _conv2d_ranges(d, 1, dy, dy.device, s, "conv2d dy") if y is None else (None, None)

            def run_dgrad():
                if rout is not None:
                    L.lib.rh_x6_set_ranges(None, L.ptr(rin), L.ptr(rout), None)
                return L.lib.rh_conv2d_bwd_data_f32(dref, L.ptr(dy), L.ptr(y), L.ptr(wp_b), L.ptr(dx), s)

            L.check(_launch2("conv2d_dgrad", d, run_dgrad), "conv2d_bwd_data")
            _attach_range(dx, rout)
        if ctx.needs_input_grad[1] or need_b:
            dw = torch.empty(ctx.wshape, device=dy.device, dtype=torch.float32)
            if need_b:
                db = torch.empty(d.c_out, device=dy.device, dtype=torch.float32)
            nbytes = L.lib.rh_conv2d_workspace_bytes(dref)
            ws = torch.empty(max(nbytes, 4) // 4, device=dy.device, dtype=torch.float32)
            # the f16 weight-gradient kernel (wgrad2d_x6.hip) scales both operands by their tensors' ranges
            use_r = _ranges_on() and y is None and L.lib.rh_conv2d_bwd_weight_kernel_family(dref) == 1
            rdy = _range_of(dy, s, "conv2d wgrad dy") if use_r else None
            rx = _range_of(x, s, "conv2d wgrad x") if use_r else None

            def run_wgrad():
                if use_r:
                    L.lib.rh_x6_set_ranges(L.ptr(rdy), L.ptr(rx), None, None)
                return L.lib.rh_conv2d_bwd_weight_f32(dref, L.ptr(dy), L.ptr(y), L.ptr(x), L.ptr(dw), L.ptr(db) if need_b else None,
                                                      L.ptr(ws), nbytes, s)

            L.check(_launch2("conv2d_wgrad", d, run_wgrad), "conv2d_bwd_weight")
        return dx, dw, db, None, None, None, None, None


def conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, stride=1, padding=0, dilation=1,
           act: int = ACT_NONE, slope: float = 0.2) -> Tensor:
    """``act(F.conv2d(x, weight, bias, stride, padding, dilation))`` on the HIP kernels (groups == 1)."""
    return _Conv2dFn.apply(x, weight, bias, stride, padding, dilation, act, float(slope))


# --------------------------------------------------------------------------- PQMF
# ``fold`` = (tab, lpad): the folded fast form (rh_pqmf_fold_k{1,2}_f32) instead of the direct-form MFMA kernels; the
# caller (rave_amd/pqmf.py) only passes it while the stored bank equals the closed-form cosine-modulated one.
class _PqmfAnalysisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, pad: Tuple[int, int], fold=None):
        x = _chk(x, "x"); w = _chk(w, "forward_conv.weight")
        rows, one, t = x.shape
        if one != 1:
            raise RuntimeError("pqmf analysis expects (B*C, 1, T)")
        m, _, k = w.shape
        n_frames = (t + pad[0] + pad[1] - k) // m + 1
        y = torch.empty(rows, m, n_frames, device=x.device, dtype=torch.float32)
        if fold is not None:
            tab, lpad = fold
            ry = _new_range(x.device) if _ranges_on() else None       # the kernel leaves max |y| for the encoder's first conv
            if ry is not None:
                L.lib.rh_x6_set_ranges(None, None, L.ptr(ry), None)
            L.check(L.lib.rh_pqmf_fold_k1_f32(L.ptr(x), L.ptr(tab), rows, t, n_frames, lpad - pad[0], 1.0, L.ptr(y), L.stream()),
                    "pqmf_fold_k1")
            _attach_range(y, ry)
        else:
            L.check(L.lib.rh_pqmf_analysis_fwd_f32(L.ptr(x), L.ptr(w), rows, t, m, k, pad[0], n_frames, L.ptr(y), L.stream()),
                    "pqmf_analysis_fwd")
        ctx.save_for_backward(w)
        ctx.geo = (rows, t, m, k, pad[0], n_frames)
        ctx.fold = fold
        return y

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        rows, t, m, k, pl, n_frames = ctx.geo
        dy = _chk(dy, "dy")
        dx = torch.empty(rows, 1, t, device=dy.device, dtype=torch.float32)
        if ctx.fold is not None:
            tab, lpad = ctx.fold
            L.check(L.lib.rh_pqmf_fold_k2_f32(L.ptr(dy), L.ptr(tab), rows, n_frames, t, pl - lpad, 1.0, L.ptr(dx), L.stream()),
                    "pqmf_fold_k2")
        else:
            L.check(L.lib.rh_pqmf_analysis_bwd_f32(L.ptr(dy), L.ptr(w), rows, t, m, k, pl, n_frames, L.ptr(dx), L.stream()),
                    "pqmf_analysis_bwd")
        return dx, None, None, None


class _PqmfSynthesisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, w, pad: Tuple[int, int], fold=None):
        y = _chk(y, "y"); w = _chk(w, "inverse_conv.weight")
        rows, m, n_frames = y.shape
        k = w.shape[2]
        n_out = n_frames + pad[0] + pad[1] - k + 1
        x = torch.empty(rows, 1, n_out * m, device=y.device, dtype=torch.float32)
        if fold is not None:
            tab, lpad = fold
            L.check(L.lib.rh_pqmf_fold_k2_f32(L.ptr(y), L.ptr(tab), rows, n_frames, n_out * m, 496 - 16 * pad[0] - lpad, 16.0,
                                              L.ptr(x), L.stream()), "pqmf_fold_k2")
        else:
            L.check(L.lib.rh_pqmf_synthesis_fwd_f32(L.ptr(y), L.ptr(w), rows, n_frames, m, k, pad[0], n_out, L.ptr(x), L.stream()),
                    "pqmf_synthesis_fwd")
        ctx.save_for_backward(w)
        ctx.geo = (rows, n_frames, m, k, pad[0], n_out)
        ctx.fold = fold
        return x

    @staticmethod
    def backward(ctx, dx):
        (w,) = ctx.saved_tensors
        rows, n_frames, m, k, pl, n_out = ctx.geo
        dx = _chk(dx, "dx")
        dy = torch.empty(rows, m, n_frames, device=dx.device, dtype=torch.float32)
        if ctx.fold is not None:
            tab, lpad = ctx.fold
            L.check(L.lib.rh_pqmf_fold_k1_f32(L.ptr(dx), L.ptr(tab), rows, n_out * m, n_frames, lpad - (496 - 16 * pl), 16.0,
                                              L.ptr(dy), L.stream()), "pqmf_fold_k1")
        else:
            L.check(L.lib.rh_pqmf_synthesis_bwd_f32(L.ptr(dx), L.ptr(w), rows, n_frames, m, k, pl, n_out, L.ptr(dy), L.stream()),
                    "pqmf_synthesis_bwd")
        return dy, None, None, None


def pqmf_analysis(x: Tensor, w: Tensor, pad: Tuple[int, int], fold=None) -> Tensor:
    return _PqmfAnalysisFn.apply(x, w, pad, fold)


def pqmf_synthesis(y: Tensor, w: Tensor, pad: Tuple[int, int], fold=None) -> Tensor:
    return _PqmfSynthesisFn.apply(y, w, pad, fold)


# --------------------------------------------------------------------------- small ops
class _AmpTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "x")
        b, c2, l = x.shape
        y = torch.empty(b, c2 // 2, l, device=x.device, dtype=torch.float32)
        L.check(L.lib.rh_amp_tanh_fwd_f32(L.ptr(x), b, c2 // 2, l, L.ptr(y), L.stream()), "amp_tanh_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _chk(dy, "dy")
        b, c2, l = x.shape
        dx = torch.empty_like(x)
        L.check(L.lib.rh_amp_tanh_bwd_f32(L.ptr(dy), L.ptr(x), b, c2 // 2, l, L.ptr(dx), L.stream()), "amp_tanh_bwd")
        return dx


def amp_tanh(x: Tensor) -> Tensor:
    """tanh(a * sigmoid(m)) with [a | m] = x.split(C/2, 1)  (rave/blocks.py:705-711)."""
    return _AmpTanhFn.apply(x)


class _SnakeFn(torch.autograd.Function):
    """x + sin^2(alpha x)/(alpha + 1e-9), alpha (C,1) learnable (rave/blocks.py:852-860)."""

    @staticmethod
    def forward(ctx, x, alpha):
        x = _chk(x, "x"); alpha = _chk(alpha, "alpha")
        b, c, l = x.shape
        y = torch.empty_like(x)
        L.check(L.lib.rh_act_fwd_f32(L.ptr(x), L.ptr(alpha), ACT_SNAKE, 0.0, b, c, l, L.ptr(y), L.stream()), "snake_fwd")
        ctx.save_for_backward(x, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        dy = _chk(dy, "dy")
        b, c, l = x.shape
        dx = torch.empty_like(x)
        da = torch.empty_like(alpha)
        nbytes = L.lib.rh_snake_bwd_workspace_bytes(b, c)
        ws = torch.empty(max(nbytes, 4) // 4, device=x.device, dtype=torch.float32)
        L.check(L.lib.rh_snake_bwd_f32(L.ptr(dy), L.ptr(x), L.ptr(alpha), b, c, l, L.ptr(dx), L.ptr(da), L.ptr(ws),
                                       nbytes, L.stream()), "snake_bwd")
        return dx, da


def snake(x: Tensor, alpha: Tensor) -> Tensor:
    return _SnakeFn.apply(x, alpha)


class _AdainTransferFn(torch.autograd.Function):
    """(x - mean_x) / (std_x + 1e-5) * std_y + mean_y with per-(batch item, channel) statistics (rave/blocks.py:887-895).
    Backward = the same kernel on dy with zero means (d/dx = std_y / (std_x + 1e-5); the statistics are buffers)."""

    @staticmethod
    def forward(ctx, x, mean_x, std_x, mean_y, std_y):
        x = _chk(x, "x")
        b, c, l = x.shape
        for t in (mean_x, std_x, mean_y, std_y):     # the kernel walks b * c rows of each: refuse what it would read out of bounds
            if t.dim() < 2 or b > t.shape[0] or t.shape[1] != c or t[:b].numel() != b * c:
                raise ValueError(f"adain statistics {tuple(t.shape)} do not cover a batch of {b} x {c} channels")
        stats = [_chk(t[:b], "statistics") for t in (mean_x, std_x, mean_y, std_y)]
        y = torch.empty_like(x)
        L.check(L.lib.rh_adain_transfer_f32(L.ptr(x), b * c, l, *[L.ptr(t) for t in stats], L.ptr(y), L.stream()), "adain_transfer")
        ctx.save_for_backward(stats[1], stats[3])
        return y

    @staticmethod
    def backward(ctx, dy):
        std_x, std_y = ctx.saved_tensors
        dy = _chk(dy, "dy")
        b, c, l = dy.shape
        zero = torch.zeros_like(std_x)
        dx = torch.empty_like(dy)
        L.check(L.lib.rh_adain_transfer_f32(L.ptr(dy), b * c, l, L.ptr(zero), L.ptr(std_x), L.ptr(zero), L.ptr(std_y), L.ptr(dx),
                                            L.stream()), "adain_transfer_bwd")
        return dx, None, None, None, None


def adain_transfer(x: Tensor, mean_x: Tensor, std_x: Tensor, mean_y: Tensor, std_y: Tensor) -> Tensor:
    return _AdainTransferFn.apply(x, mean_x, std_x, mean_y, std_y)


def adain_stats_update(x: Tensor, mean_buf: Tensor, std_buf: Tensor, num_updates: Tensor) -> None:
    """mean_buf[:B] / std_buf[:B] <- running average with the per-row mean / unbiased std of x (B, C, L), in place
    (rave/blocks.py:876-879, 904-909); ``num_updates`` is the module's 1-element device buffer."""
    x = _chk(x.detach(), "x")
    b, c, l = x.shape
    if b > mean_buf.shape[0] or mean_buf.shape[1] != c or not (mean_buf.is_contiguous() and std_buf.is_contiguous()):
        raise ValueError(f"adain statistics buffers {tuple(mean_buf.shape)} do not cover a batch of {b} x {c} channels")
    L.check(L.lib.rh_adain_stats_update_f32(L.ptr(x), b * c, l, L.ptr(num_updates), L.ptr(mean_buf), L.ptr(std_buf), L.stream()),
            "adain_stats_update")


class _StftFrameFn(torch.autograd.Function):
    """(rows, T) -> (rows, T//hop + 1, n_fft): centre + reflect pad + window, and the adjoint."""

    @staticmethod
    def forward(ctx, x, window, n_fft: int, hop: int):
        x = _chk(x, "x"); window = _chk(window, "window")
        rows, t = x.shape
        nf = t // hop + 1
        frames = torch.empty(rows, nf, n_fft, device=x.device, dtype=torch.float32)
        L.check(L.lib.rh_stft_frame_fwd_f32(L.ptr(x), L.ptr(window), rows, t, n_fft, hop, nf, L.ptr(frames), L.stream()),
                "stft_frame_fwd")
        ctx.save_for_backward(window)
        ctx.geo = (rows, t, n_fft, hop, nf)
        return frames

    @staticmethod
    def backward(ctx, dfr):
        (window,) = ctx.saved_tensors
        rows, t, n_fft, hop, nf = ctx.geo
        dfr = _chk(dfr, "dframes")
        dx = torch.empty(rows, t, device=dfr.device, dtype=torch.float32)
        L.check(L.lib.rh_stft_frame_bwd_f32(L.ptr(dfr), L.ptr(window), rows, t, n_fft, hop, nf, L.ptr(dx), L.stream()),
                "stft_frame_bwd")
        return dx, None, None, None


def stft_frames(x: Tensor, window: Tensor, n_fft: int, hop: int) -> Tensor:
    return _StftFrameFn.apply(x, window, n_fft, hop)


class _SpectralDistanceFn(torch.autograd.Function):
    """mean((|Sx|-|Sy|)^2)/mean(|Sx|^2) + mean(|log(|Sx|+eps) - log(|Sy|+eps)|) on complex STFTs."""

    @staticmethod
    def forward(ctx, sx, sy, eps: float):
        if not (sx.is_cuda and sy.is_cuda and sx.is_complex() and sy.is_complex() and sx.shape == sy.shape):
            raise RuntimeError("rave_amd spectral_distance: expects two complex GPU tensors of equal shape")
        sx = sx.contiguous(); sy = sy.contiguous()
        n = sx.numel()
        rx, ry = torch.view_as_real(sx), torch.view_as_real(sy)
        sums = torch.empty(3, device=sx.device, dtype=torch.float32)
        nbytes = L.lib.rh_spectral_distance_workspace_bytes()
        ws = torch.empty(nbytes // 4, device=sx.device, dtype=torch.float32)
        L.check(L.lib.rh_spectral_distance_fwd_f32(L.ptr(rx), L.ptr(ry), n, eps, L.ptr(sums), L.ptr(ws), nbytes,
                                                   L.stream()), "spectral_distance_fwd")
        ctx.save_for_backward(sx, sy, sums)
        ctx.eps = eps
        return sums[0] / sums[1] + sums[2] / n

    @staticmethod
    def backward(ctx, g):
        sx, sy, sums = ctx.saved_tensors
        g = g.contiguous().reshape(1).float()
        dsx = torch.empty_like(sx) if ctx.needs_input_grad[0] else None
        dsy = torch.empty_like(sy) if ctx.needs_input_grad[1] else None
        L.check(L.lib.rh_spectral_distance_bwd_f32(
            L.ptr(torch.view_as_real(sx)), L.ptr(torch.view_as_real(sy)), L.ptr(sums), L.ptr(g), sx.numel(), ctx.eps,
            None if dsx is None else torch.view_as_real(dsx).data_ptr(),
            None if dsy is None else torch.view_as_real(dsy).data_ptr(), 0, L.stream()), "spectral_distance_bwd")
        return dsx, dsy, None


def spectral_distance(sx: Tensor, sy: Tensor, eps: float) -> Tensor:
    return _SpectralDistanceFn.apply(sx, sy, eps)


class _StftDistanceFn(torch.autograd.Function):
    """AudioDistanceV1 on one STFT scale from the windowed FRAMES (rows, n_frames, n_fft) of both signals:
    rfft (rocFFT) + fused distance kernel forward; backward = fused gradient kernel emitting the operand of the
    C2R adjoint of rfft + one unnormalised irfft per signal (instead of autograd's FftR2CBackward: complex
    zero-fill + copy + C2C + real part)."""

    @staticmethod
    def forward(ctx, fx, fy, eps: float):
        fx = _chk(fx, "frames_x"); fy = _chk(fy, "frames_y")
        if fx.shape != fy.shape:
            raise RuntimeError("rave_amd stft_distance: frame tensors differ in shape")
        n_fft = fx.shape[-1]
        from . import fft as F
        # the frames are scratch produced by the framing kernel (its backward only needs the window): the FFT
        # may overwrite them
        sx = F.rfft_last(fx)
        sy = F.rfft_last(fy)
        n = sx.numel()
        sums = torch.empty(3, device=fx.device, dtype=torch.float32)
        nbytes = L.lib.rh_spectral_distance_workspace_bytes()
        ws = torch.empty(nbytes // 4, device=fx.device, dtype=torch.float32)
        L.check(L.lib.rh_spectral_distance_fwd_f32(L.ptr(torch.view_as_real(sx)), L.ptr(torch.view_as_real(sy)), n, eps,
                                                   L.ptr(sums), L.ptr(ws), nbytes, L.stream()), "spectral_distance_fwd")
        ctx.save_for_backward(sx, sy, sums)
        ctx.eps, ctx.n_fft = eps, n_fft
        return sums[0] / sums[1] + sums[2] / n

    @staticmethod
    def backward(ctx, g):
        sx, sy, sums = ctx.saved_tensors
        g = g.contiguous().reshape(1).float()
        hx = torch.empty_like(sx) if ctx.needs_input_grad[0] else None
        hy = torch.empty_like(sy) if ctx.needs_input_grad[1] else None
        L.check(L.lib.rh_spectral_distance_bwd_f32(
            L.ptr(torch.view_as_real(sx)), L.ptr(torch.view_as_real(sy)), L.ptr(sums), L.ptr(g), sx.numel(), ctx.eps,
            None if hx is None else torch.view_as_real(hx).data_ptr(),
            None if hy is None else torch.view_as_real(hy).data_ptr(), sx.shape[-1], L.stream()), "spectral_distance_bwd")
        from . import fft as F
        dfx = F.irfft_last_unnormalized(hx, ctx.n_fft) if hx is not None else None
        dfy = F.irfft_last_unnormalized(hy, ctx.n_fft) if hy is not None else None
        return dfx, dfy, None


def stft_distance(frames_x: Tensor, frames_y: Tensor, eps: float) -> Tensor:
    """CONTRACT: the frame tensors are SCRATCH -- the in-place hipFFT R2C call may overwrite them.  Pass tensors
    nothing else reads afterwards (``ops.stft_frames`` output, as rave_amd.losses does) or clones."""
    return _StftDistanceFn.apply(frames_x, frames_y, eps)


def _stft_one_finalize() -> bool:
    import os
    return os.environ.get("RH_STFT_ONE_FINALIZE", "1") != "0"


class _MultiScaleStftDistanceFn(torch.autograd.Function):
    """AudioDistanceV1 over ALL scales of MultiScaleSTFT as one autograd node (rave/core.py:269-344): per scale the HIP
    framing kernel, rocFFT R2C and the fused distance kernel; the sum over the scales in one tiny launch; backward: per
    scale the fused gradient kernel, the C2R adjoint and the framing adjoint ACCUMULATING into one dx / dy buffer.  The
    per-scale nodes of `stft_distance` cost ~10 scalar ATen launches per scale (divisions, additions and their
    backwards) and 4 full-size gradient additions per signal -- ~30 % of the launches of a VAE-phase step."""

    @staticmethod
    def forward(ctx, x, y, eps: float, scales, *windows):
        x = _chk(x, "x"); y = _chk(y, "y")
        if x.shape != y.shape or x.dim() != 2:
            raise RuntimeError("rave_amd multiscale_stft_distance: expects two (rows, T) tensors of equal shape")
        from . import fft as F
        rows, t = x.shape
        s = L.stream()
        dev = x.device
        ns = len(scales)
        sums = torch.empty(ns, 3, device=dev, dtype=torch.float32)
        if _stft_fused_ok(scales, t, rows):
            # transform inside the kernel (stft_loss.hip): nothing but the two waveforms is read, nothing but 3 sums per
            # scale written; the backward recomputes the spectra from the saved waveforms
            inv_n = []
            one_finalize = ns <= 8 and _stft_one_finalize()
            parts = []
            for i, (n_fft, win) in enumerate(zip(scales, windows)):
                win = _chk(win, "window")
                nbytes = L.lib.rh_stft_loss_workspace_bytes(n_fft, t, rows)
                ws = torch.empty(max(nbytes // 4, 1), device=dev, dtype=torch.float32)
                L.check(L.lib.rh_stft_loss_fwd_f32(L.ptr(x), L.ptr(y), L.ptr(win), L.ptr(_twiddle(n_fft, dev)), rows, t, n_fft, eps,
                                                   None if one_finalize else sums[i].data_ptr(), L.ptr(ws), nbytes, s), "stft_loss_fwd")
                parts.append((ws, nbytes))
                inv_n.append(1.0 / (rows * (t // (n_fft // 4) + 1) * (n_fft // 2 + 1)))
            key = (tuple(inv_n), str(dev))
            if key not in _INV_N and not torch.cuda.is_current_stream_capturing():
                _INV_N[key] = torch.tensor(inv_n, dtype=torch.float32, device=dev)
            inv = _inv_n_cached(tuple(inv_n), dev)
            out = torch.empty((), device=dev, dtype=torch.float32)
            if one_finalize:       # the ordered finalize of every scale + the sum over the scales: one launch
                pp = (C.c_void_p * ns)(*[w.data_ptr() for w, _ in parts])
                nb = (C.c_int64 * ns)(*[b for _, b in parts])
                L.check(L.lib.rh_stft_loss_finalize_all_f32(pp, nb, ns, L.ptr(inv), L.ptr(sums), L.ptr(out), s), "stft_loss_finalize_all")
            else:
                L.check(L.lib.rh_spectral_total_f32(L.ptr(sums), L.ptr(inv), ns, L.ptr(out), s), "spectral_total")
            ctx.save_for_backward(sums, *windows, x, y)
            ctx.meta = (rows, t, float(eps), tuple(int(v) for v in scales))
            ctx.fused = True
            return out
        ctx.fused = False
        nbytes = L.lib.rh_spectral_distance_workspace_bytes()
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        saved = []
        inv_n = []
        for i, (n_fft, win) in enumerate(zip(scales, windows)):
            win = _chk(win, "window")
            hop = n_fft // 4
            nf = t // hop + 1
            # both signals in one frames buffer: ONE batched R2C per scale
            fr = torch.empty(2, rows, nf, n_fft, device=dev, dtype=torch.float32)
            for k, sig in enumerate((x, y)):
                L.check(L.lib.rh_stft_frame_fwd_f32(L.ptr(sig), L.ptr(win), rows, t, n_fft, hop, nf, fr[k].data_ptr(), s), "stft_frame_fwd")
            spec = F.rfft_last(fr)
            sx, sy = spec[0], spec[1]
            n = sx.numel()
            L.check(L.lib.rh_spectral_distance_fwd_f32(L.ptr(torch.view_as_real(sx)), L.ptr(torch.view_as_real(sy)), n, eps,
                                                       sums[i].data_ptr(), L.ptr(ws), nbytes, s), "spectral_distance_fwd")
            saved += [sx, sy]
            inv_n.append(1.0 / n)
        key = (tuple(inv_n), str(dev))
        if key not in _INV_N and not torch.cuda.is_current_stream_capturing():
            _INV_N[key] = torch.tensor(inv_n, dtype=torch.float32, device=dev)
        inv = _inv_n_cached(tuple(inv_n), dev)
        out = torch.empty((), device=dev, dtype=torch.float32)
        L.check(L.lib.rh_spectral_total_f32(L.ptr(sums), L.ptr(inv), ns, L.ptr(out), s), "spectral_total")
        ctx.save_for_backward(sums, *windows, *saved)
        ctx.meta = (rows, t, float(eps), tuple(int(v) for v in scales))
        return out

    @staticmethod
    def backward(ctx, g):
        rows, t, eps, scales = ctx.meta
        ns = len(scales)
        sums = ctx.saved_tensors[0]
        windows = ctx.saved_tensors[1:1 + ns]
        specs = ctx.saved_tensors[1 + ns:]
        from . import fft as F
        g = g.contiguous().reshape(1).float()
        s = L.stream()
        dev = g.device
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        outs = [torch.empty(rows, t, device=dev, dtype=torch.float32) if nd else None for nd in need]
        if ctx.fused:
            x, y = specs
            for i, n_fft in enumerate(scales):
                L.check(L.lib.rh_stft_loss_bwd_f32(L.ptr(x), L.ptr(y), L.ptr(windows[i]), L.ptr(_twiddle(n_fft, dev)), rows, t, n_fft,
                                                   eps, sums[i].data_ptr(), L.ptr(g), None if outs[0] is None else L.ptr(outs[0]),
                                                   None if outs[1] is None else L.ptr(outs[1]), 1 if i > 0 else 0, s), "stft_loss_bwd")
            return (outs[0], outs[1], None, None) + (None,) * ns
        for i, n_fft in enumerate(scales):
            sx, sy = specs[2 * i], specs[2 * i + 1]
            hop = n_fft // 4
            nf = t // hop + 1
            both = need[0] and need[1]
            hh = torch.empty((2,) + tuple(sx.shape), device=dev, dtype=sx.dtype) if both else None
            hx = hh[0] if both else (torch.empty_like(sx) if need[0] else None)
            hy = hh[1] if both else (torch.empty_like(sy) if need[1] else None)
            L.check(L.lib.rh_spectral_distance_bwd_f32(
                L.ptr(torch.view_as_real(sx)), L.ptr(torch.view_as_real(sy)), sums[i].data_ptr(), L.ptr(g), sx.numel(), eps,
                None if hx is None else torch.view_as_real(hx).data_ptr(),
                None if hy is None else torch.view_as_real(hy).data_ptr(), sx.shape[-1], s), "spectral_distance_bwd")
            if both:        # ONE batched C2R for both signals
                dfr = F.irfft_last_unnormalized(hh, n_fft)
                parts = ((dfr[0], outs[0]), (dfr[1], outs[1]))
            else:
                parts = tuple((F.irfft_last_unnormalized(h, n_fft), o) for h, o in ((hx, outs[0]), (hy, outs[1])) if h is not None)
            for d_fr, o in parts:
                L.check(L.lib.rh_stft_frame_bwd_acc_f32(d_fr.data_ptr(), L.ptr(windows[i]), rows, t, n_fft, hop, nf, L.ptr(o),
                                                        1 if i > 0 else 0, s), "stft_frame_bwd")
        return (outs[0], outs[1], None, None) + (None,) * ns


_INV_N = {}
_TWIDDLE = {}


def _twiddle(n: int, dev) -> Tensor:
    """e^{-2 pi i k / n}, k < n, as (n, 2) f32 (rounded from f64), made once per size and device outside any capture."""
    k = (int(n), str(dev))
    t = _TWIDDLE.get(k)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("rave_amd multiscale_stft_distance: first call inside a hipGraph capture (run one eager step first)")
        import math
        a = torch.arange(n, dtype=torch.float64) * (-2.0 * math.pi / n)
        t = _TWIDDLE[k] = torch.stack([torch.cos(a), torch.sin(a)], -1).to(torch.float32).to(dev).contiguous()
    return t


def _stft_fused_ok(scales, t: int, rows: int) -> bool:
    import os
    if os.environ.get("RH_STFT_FUSED", "1") == "0":      # 0: framing kernels + rocFFT + spectral kernels (the older path)
        return False
    return all(L.lib.rh_stft_loss_supported(int(n), int(n) // 4, int(t), int(rows)) for n in scales)


def _inv_n_cached(key, dev):
    """1/n per scale on the device, created outside any capture (a hipGraph cannot contain the host-to-device copy)."""
    k = (key, str(dev))
    if k not in _INV_N:
        raise RuntimeError("rave_amd multiscale_stft_distance: first call inside a hipGraph capture (run one eager step first)")
    return _INV_N[k]


def multiscale_stft_distance(x: Tensor, y: Tensor, windows, scales, eps: float) -> Tensor:
    """sum over the scales of mean((|Sx|-|Sy|)^2)/mean(|Sx|^2) + mean(|log(|Sx|+eps) - log(|Sy|+eps)|) for (rows, T) signals.
"""
    return _MultiScaleStftDistanceFn.apply(x, y, float(eps), tuple(int(s) for s in scales), *windows)


# --------------------------------------------------------------------------- reparametrisation + KL
class _ReparamFn(torch.autograd.Function):
    """VariationalEncoder.reparametrize (rave/blocks.py:727-745) on rh_reparam_{fwd,bwd}_f32: (zs, kl) from z = [mean | scale]
    and the noise draw eps."""

    @staticmethod
    def forward(ctx, z, eps):
        z = _chk(z, "z"); eps = _chk(eps, "eps")
        b, c2, l = z.shape
        c = c2 // 2
        if c2 % 2 or tuple(eps.shape) != (b, c, l):
            raise RuntimeError("rave_amd reparametrize: z must be (B, 2C, L) and eps (B, C, L)")
        zs = torch.empty(b, c, l, device=z.device, dtype=torch.float32)
        kl = torch.empty((), device=z.device, dtype=torch.float32)
        nbytes = L.lib.rh_reparam_workspace_bytes()
        ws = torch.empty(nbytes // 4, device=z.device, dtype=torch.float32)
        rz = _new_range(z.device) if _ranges_on() else None           # the kernel leaves max |zs| for the decoder's first conv
        if rz is not None:
            L.lib.rh_x6_set_ranges(None, None, L.ptr(rz), None)
        L.check(L.lib.rh_reparam_fwd_f32(L.ptr(z), L.ptr(eps), b, c, l, L.ptr(zs), L.ptr(kl), L.ptr(ws), nbytes, L.stream()),
                "reparam_fwd")
        _attach_range(zs, rz)
        ctx.save_for_backward(z, eps)
        return zs, kl

    @staticmethod
    def backward(ctx, dzs, dkl):
        z, eps = ctx.saved_tensors
        b, c2, l = z.shape
        dzs = _chk(dzs, "dzs")
        dkl = None if dkl is None else dkl.contiguous().reshape(1).float()
        dz = torch.empty_like(z)
        L.check(L.lib.rh_reparam_bwd_f32(L.ptr(z), L.ptr(eps), L.ptr(dzs), L.ptr(dkl), b, c2 // 2, l, L.ptr(dz), L.stream()),
                "reparam_bwd")
        return dz, None


def reparametrize(z: Tensor, eps: Tensor):
    """(eps * std + mean, KL) of rave/blocks.py:727-745 for z = [mean | scale] (B, 2C, L)."""
    return _ReparamFn.apply(z, eps)


# --------------------------------------------------------------------------- feature matching (GAN phase)
def _dense_batch_major(t: Tensor) -> bool:
    """True if ``t``'s memory is one dense block in which dim 0 is outermost (any order of the other dims): its first
    half along dim 0 is then the first half of the block -- the period discriminators hand out permuted views."""
    if t.dim() < 1 or t.shape[0] % 2 or t.numel() == 0:
        return False
    if t.is_contiguous():
        return True
    n = t.numel()
    if t.stride(0) * t.shape[0] != n:
        return False
    # the remaining dims must tile [0, stride(0)) exactly: sort by stride and check the products
    dims = sorted(((t.stride(i), t.shape[i]) for i in range(1, t.dim()) if t.shape[i] > 1))
    acc = 1
    for st, sz in dims:
        if st != acc:
            return False
        acc *= sz
    return acc == t.stride(0)


class _FeatureMatchingFn(torch.autograd.Function):
    """sum_i w_i * mean_difference(real_i, fake_i, "L1"[, relative]) over UNSPLIT feature maps (real half of the batch first),
    rh_feature_matching_{fwd,bwd}_f32: one pass per direction over every feature map instead of ~10 ATen passes."""

    @staticmethod
    def forward(ctx, relative: bool, weights, *feats):
        fs = []
        for f in feats:
            if not f.is_cuda or f.dtype != torch.float32:
                raise RuntimeError("rave_amd feature_matching: feature maps must be float32 tensors on the GPU")
            # NOT _chk(): the period discriminators hand out permuted views of dense period-major tensors; they are used
            # (and their gradients written) in their own memory order -- no copy, and the gradient goes back through the
            # view without one
            fs.append(f if _dense_batch_major(f) else f.contiguous())
        n = len(fs)
        dev = fs[0].device
        arr = (L.FmItem * n)()
        for i, (f, w) in enumerate(zip(fs, weights)):
            half = f.numel() // 2
            arr[i].f, arr[i].df, arr[i].half = f.data_ptr(), None, half
            arr[i].w = float(w) if relative else float(w) / half
        nbytes = L.lib.rh_feature_matching_workspace_bytes(arr, n)
        if nbytes < 0:
            raise RuntimeError("rave_amd feature_matching: too many feature maps for one call")
        ws = torch.empty(max(nbytes // 4, 1), device=dev, dtype=torch.float32)
        sums = torch.empty(n, 2, device=dev, dtype=torch.float32)
        out = torch.empty((), device=dev, dtype=torch.float32)
        L.check(L.lib.rh_feature_matching_fwd_f32(arr, n, int(relative), L.ptr(ws), nbytes, L.ptr(sums), L.ptr(out), L.stream()),
                "feature_matching_fwd")
        ctx.save_for_backward(sums, *fs)
        ctx.meta = (bool(relative), [arr[i].w for i in range(n)])
        return out

    @staticmethod
    def backward(ctx, g):
        relative, ws = ctx.meta
        sums = ctx.saved_tensors[0]
        fs = ctx.saved_tensors[1:]
        n = len(fs)
        g = g.contiguous().reshape(1).float()
        arr = (L.FmItem * n)()
        dfs = []
        for i, f in enumerate(fs):
            df = torch.empty_like(f)           # same strides as f (dense): element k of the block is element k of f's block
            dfs.append(df)
            arr[i].f, arr[i].df, arr[i].half, arr[i].w = f.data_ptr(), df.data_ptr(), f.numel() // 2, ws[i]
        L.check(L.lib.rh_feature_matching_bwd_f32(arr, n, int(relative), L.ptr(sums), L.ptr(g), L.stream()), "feature_matching_bwd")
        return (None, None) + tuple(dfs)


def feature_matching(features, weights, relative: bool) -> Tensor:
    """``features``: unsplit discriminator feature maps (2B, ...), real half first; ``weights``: one factor per map (the
    1 / (maps of its discriminator x discriminators) of rave/model.py:359-372)."""
    return _FeatureMatchingFn.apply(bool(relative), tuple(float(w) for w in weights), *features)


class _LossCombineFn(torch.autograd.Function):
    """(total, scaled): scaled_i = w1_i * value_i, total = sum_i scaled_i * w2_i -- rh_loss_combine_fwd/bwd_f32."""

    @staticmethod
    def forward(ctx, meta, *vals):
        w1, w1_dev, w2 = meta
        n = len(vals)
        vals = [_chk(v, "loss term") for v in vals]
        if any(v.numel() != 1 for v in vals):
            raise RuntimeError("rave_amd loss_combine: every term must be a scalar tensor")
        dev = vals[0].device
        arr = (L.LossItem * n)()
        for a, v, a1, d1, a2 in zip(arr, vals, w1, w1_dev, w2):
            a.value, a.w1_dev, a.w1, a.w2 = L.ptr(v), (L.ptr(d1) if d1 is not None else None), float(a1), float(a2)
        scaled = torch.empty(n, device=dev, dtype=torch.float32)
        total = torch.empty((), device=dev, dtype=torch.float32)
        L.check(L.lib.rh_loss_combine_fwd_f32(arr, n, L.ptr(scaled), L.ptr(total), L.stream()), "loss_combine_fwd")
        ctx.arr, ctx.n = arr, n
        ctx.keep = (vals, [d for d in w1_dev if d is not None])        # (the table holds raw pointers)
        ctx.mark_non_differentiable(scaled)
        return total, scaled

    @staticmethod
    def backward(ctx, g, _g_scaled):
        g = _chk(g, "grad").reshape(1)
        grads = torch.empty(ctx.n, device=g.device, dtype=torch.float32)
        L.check(L.lib.rh_loss_combine_bwd_f32(ctx.arr, ctx.n, L.ptr(g), L.ptr(grads), L.stream()), "loss_combine_bwd")
        return (None,) + tuple(grads[i].reshape(v.shape) if ctx.needs_input_grad[i + 1] else None
                               for i, v in enumerate(ctx.keep[0]))


def loss_combine(values, w1, w2, w1_dev=None):
    """The generator loss of a step in one launch each way (rave/model.py:336-344, 392-412): ``scaled[i] = w1[i] * values[i]``
    (or ``values[i] * w1_dev[i]`` where a device scalar is given -- the captured step's beta_factor) and
    ``total = sum_i scaled[i] * w2[i]`` in term order, every product rounded on its own: the bits of the reference's
    ``weights[k] * v`` ... ``loss_gen_value += v * self.weights.get(k, 1.)`` chain.  Returns (total, scaled); ``scaled`` is for
    logging (not differentiable)."""
    n = len(values)
    w1_dev = list(w1_dev) if w1_dev is not None else [None] * n
    return _LossCombineFn.apply((tuple(float(a) for a in w1), tuple(w1_dev), tuple(float(a) for a in w2)), *values)


class _AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "x")
        l_in = x.shape[-1]
        rows = x.numel() // max(l_in, 1)
        y = torch.empty(*x.shape[:-1], l_in // 2, device=x.device, dtype=torch.float32)
        L.check(L.lib.rh_avgpool2_fwd_f32(L.ptr(x), rows, l_in, L.ptr(y), L.stream()), "avgpool2_fwd")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _chk(dy, "dy")
        l_in = ctx.shape[-1]
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        rows = dx.numel() // max(l_in, 1)
        L.check(L.lib.rh_avgpool2_bwd_f32(L.ptr(dy), rows, l_in, L.ptr(dx), L.stream()), "avgpool2_bwd")
        return dx


def avg_pool2(x: Tensor) -> Tensor:
    return _AvgPool2Fn.apply(x)
