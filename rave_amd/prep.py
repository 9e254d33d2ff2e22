"""Batched weight preparation: the weight norm g*v/||v|| and both MFMA-friendly packed copies of every
weight-normalised conv of a model in TWO kernel launches per training step (rh_prep_run_f32), instead
of two launches per layer.  The reference recomputes every normalised weight once per forward
(torch.nn.utils.weight_norm pre-hook, rave/blocks.py:15-22); this does the same work, batched.

Usage (rave_amd/model.py does this inside training_step)::

    prep = WeightPrep(model)      # once, after .to(device)
    prep.run()                    # start of every step: refresh all packed weights
    ... forward / backward ...    # conv modules pick up module._prepacked
    prep.release()                # end of step: later stray forwards re-pack on their own

Packed buffers are persistent (allocated once); a step's backward reads them before the next
``run()`` overwrites them (same stream).

CONTRACT: the packed buffers are rewritten through raw pointers, so autograd's version counters do not see it.
Every backward of a graph built under one ``run()`` must finish before the next ``run()``: a ``retain_graph``
backward replayed after it would read the NEW weights against the OLD activations without any error.
``training_step`` (forward, backward, optimizer step inside one prepare / release pair) satisfies this; code
that keeps graphs across steps must not use WeightPrep (the modules then repack per forward into fresh tensors).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch
import torch.nn as nn

from . import _lib as L
from . import cc
from .ops import ConvGeom, _desc


def _geom_of(m: nn.Module) -> ConvGeom:
    if isinstance(m, (cc.Conv1d, cc.ConvTranspose1d)):
        return m.geom()
    if isinstance(m, cc.PlainConv1d):
        return ConvGeom(stride=m.stride[0], dilation=m.dilation[0], pad_left=m.padding[0], pad_right=m.padding[0])
    if isinstance(m, cc.Conv2dK1):
        return ConvGeom(stride=m.stride[0], pad_left=m.padding[0], pad_right=m.padding[0])
    raise TypeError(type(m))


class WeightPrep:
    def __init__(self, model: nn.Module):
        self.mods: List[nn.Module] = [m for m in model.modules()
                                      if isinstance(m, (cc.Conv1d, cc.ConvTranspose1d, cc.PlainConv1d, cc.Conv2dK1))
                                      and getattr(m, "weight_g", None) is not None and m.groups == 1
                                      and m.weight_v.is_cuda]
        self.n = len(self.mods)
        self.bufs = []
        if self.n == 0:
            return
        isz = L.lib.rh_prep_item_bytes()
        host = (C.c_uint8 * L.lib.rh_prep_array_bytes(self.n))()      # the items + the lookup tables rh_prep_link appends
        dev = self.mods[0].weight_v.device
        for i, m in enumerate(self.mods):
            v, g = m.weight_v, m.weight_g
            k = v.shape[2]
            transposed = isinstance(m, cc.ConvTranspose1d)
            c_in, c_out = (v.shape[0], v.shape[1]) if transposed else (v.shape[1], v.shape[0])
            d = _desc(_geom_of(m), 1, c_in, c_out, 1, 1, k)
            dref = C.byref(d)
            rows = v.shape[0]
            # norms | scale | per-row max |w| | per-row sum |w| (the last two: range statistics rh_prep_run_f32 keeps behind scale)
            ns = torch.empty(4, rows, device=dev, dtype=torch.float32)
            wp_f = torch.empty(L.lib.rh_conv1d_packed_floats(dref, 0), device=dev, dtype=torch.float32)
            wp_b = torch.empty(L.lib.rh_conv1d_packed_floats(dref, 1), device=dev, dtype=torch.float32)
            item = C.cast(C.byref(host, i * isz), C.c_void_p)
            L.check(L.lib.rh_prep_fill_item(dref, L.ptr(v), L.ptr(g), L.ptr(ns[0]), L.ptr(ns[1]), L.ptr(wp_f),
                                            L.ptr(wp_b), item), "prep_fill_item")
            self.bufs.append((wp_f, wp_b, ns[0], ns, v.data_ptr(), g.data_ptr()))
        tr, tb = C.c_int64(0), C.c_int64(0)
        L.check(L.lib.rh_prep_link(C.cast(host, C.c_void_p), self.n, C.byref(tr), C.byref(tb)), "prep_link")
        self.total_rows, self.total_blocks = tr.value, tb.value
        self.items = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)

    def _versions(self):
        return tuple((m.weight_v._version, m.weight_g._version) for m in self.mods)

    def invalidate(self) -> None:
        """Forget which parameter versions the packed buffers hold.  Needed after parameters were rewritten behind
        autograd's version counters (a replayed hipGraph's optimizer step)."""
        self._packed_versions = None

    def run(self, reuse: bool = False) -> None:
        """Refresh the packed weights.  ``reuse=True`` (inference / no-grad forwards): skip the two launches when no
        parameter changed since the last refresh and re-attach the persistent buffers -- such a forward then pays the
        weight norm + repack once, not per call.  "Changed" = the version counters of every weight_v / weight_g
        (``load_state_dict``, in-place tensor ops and foreach optimizers bump them) OR an explicit ``invalidate()``:
        torch's FUSED optimizers update parameters WITHOUT touching the counters (checked: torch 2.10, Adam(fused=True)),
        so a training loop must call ``invalidate()`` after its optimizer step -- ``RAVE.training_step`` and
        ``GraphedTrainingStep`` do.  The default (``reuse=False``) always refreshes, as the reference recomputes every
        normalised weight on every forward.  While a hipGraph is being recorded the kernels always run."""
        if self.n == 0:
            return
        for m, b in zip(self.mods, self.bufs):
            if m.weight_v.data_ptr() != b[4] or m.weight_g.data_ptr() != b[5]:
                raise RuntimeError("rave_amd.WeightPrep: a parameter was re-allocated; rebuild the WeightPrep")
        capturing = torch.cuda.is_current_stream_capturing()
        ver = self._versions()
        if not reuse or capturing or ver != getattr(self, "_packed_versions", None):
            L.check(L.lib.rh_prep_run_f32(self.items.data_ptr(), self.n, self.total_rows, self.total_blocks, L.stream()),
                    "prep_run")
            # a recorded graph re-runs the repack at every replay, against parameters the replayed optimizer has
            # changed without touching the version counters: never trust the cache across a capture
            self._packed_versions = None if capturing else ver
        for m, b in zip(self.mods, self.bufs):
            m._prepacked = (b[0], b[1], b[2])

    def release(self) -> None:
        for m in self.mods:
            m._prepacked = None
