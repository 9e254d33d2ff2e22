"""Counting LeakyReLU gate flips between two evaluations of the hot path (VERDICT r3, weak #1a).

A LeakyReLU gate is the one discontinuity of the hot path's backward pass: a pre-activation within rounding of zero takes
the other slope in another fp32 (or the fp64) evaluation, which changes that element's gradient 5x (slope 0.2) or 10x
(slope 0.1) and moves every parameter gradient UPSTREAM of it (in the backward sense: every layer whose output flows
through the gate) by ~0.8 / sqrt(numel).  The end-to-end gradient tests therefore do not "tolerate a few loose tensors":
they COUNT the flips -- sign masks of the pre-activations of both evaluations -- and require

  * every parameter gradient outside the tight bound to have at least one flipped gate downstream of its layer, and
  * (the same statement read the other way) no tensor whose downstream gates all agree to be outside the tight bound.

Two sources of the pre-activations:
  * generator side (a chain): ``rave_amd.ops.gate_log_*`` records every conv launch in forward order with the ids of its
    parameters and the tensor its fused LeakyReLU reads; the CPU oracle's are recorded by wrapping
    ``torch.nn.functional.leaky_relu`` while it runs (``OracleGates``) -- same order, one entry per gate;
  * discriminators (several nets, MRD is a tree): the feature maps ARE the gates (1-D nets hand out the pre-activation,
    the 2-D nets the activated map, whose sign is the pre-activation's); which parameters lie upstream of a map is read
    from the autograd graph of the HIP run (``params_upstream_of``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class OracleGates:
    """``with OracleGates() as g: oracle(...)`` -> ``g.masks``: one bool tensor (x > 0) per F.leaky_relu call, in call order."""

    def __enter__(self):
        self.masks = []
        self._orig = F.leaky_relu

        def rec(x, *a, **kw):
            self.masks.append((x.detach() > 0))
            return self._orig(x, *a, **kw)

        F.leaky_relu = rec
        return self

    def __exit__(self, *exc):
        F.leaky_relu = self._orig
        return False


def name_gate_log(log, model):
    """ops.gate_log_end() entries -> [(parameter names of the launch, pre-activation tensor or None)]."""
    names = {id(p): k for k, p in model.named_parameters()}
    return [(tuple(names[i] for i in ids if i in names), x) for ids, x in log]


def chain_flips(log, other):
    """Flips per launch of a CHAIN (generator side).  ``log``: named gate log of the HIP run; ``other``: either the named
    gate log of a second HIP run (same launches) or the list of oracle masks (one per gated launch, forward order).
    Returns (flips per launch, total gate elements, worst |pre-activation| / rms among flipped elements)."""
    gated = [i for i, (_, x) in enumerate(log) if x is not None]
    if other and isinstance(other[0], tuple):
        assert len(other) == len(log), (len(other), len(log))
        masks = [None if x is None else (x > 0, x) for _, x in other]
        masks = [masks[i] for i in gated]
    else:
        assert len(other) == len(gated), (len(other), len(gated))
        masks = [(m, None) for m in other]
    flips = [0] * len(log)
    total = 0
    worst = 0.0
    for i, (m, xo) in zip(gated, masks):
        x = log[i][1]
        assert tuple(m.shape) == tuple(x.shape), (i, tuple(m.shape), tuple(x.shape))
        a = x > 0
        d = a != m.to(a.device)
        n = int(d.sum())
        flips[i] = n
        total += x.numel()
        if n:
            rms = float(x.double().pow(2).mean().sqrt())
            mag = x.abs()[d].max()
            if xo is not None:
                mag = torch.maximum(mag, xo.abs()[d].max())
            worst = max(worst, float(mag) / max(rms, 1e-30))
    return flips, total, worst


def flips_downstream_by_param(log, flips):
    """{parameter name: number of flipped gate elements in launches AFTER the one that carries the parameter} -- in a chain a
    gate on the input of launch j modulates the gradient that flows back to every launch i < j (the gate on a launch's
    own input only enters its weight gradient through act(x), which is continuous)."""
    after = [0] * (len(log) + 1)
    for i in range(len(log) - 1, -1, -1):
        after[i] = after[i + 1] + flips[i]
    out = {}
    for i, (names, _) in enumerate(log):
        for k in names:
            out[k] = after[i + 1]
    return out


def flip_allowance_by_param(log, flips):
    """{parameter name: the relative gradient change the flips DOWNSTREAM of it account for}.  One flipped LeakyReLU(0.2) gate
    in a map of n elements changes that element's derivative by 0.8 of its value, i.e. the gradient flowing back through
    the map by ~ 0.8 / sqrt(n) relative L2 (measured: one flip in the golden CAPACITY-96 fixture moved the two tensors upstream
    of it by 2.0e-4 and 2.6e-3); independent flips add in quadrature:
    allowance = 0.8 * sqrt(sum over later launches of flips_j / numel_j).  0 for a tensor with no flipped gate downstream."""
    term = [0.0] * (len(log) + 1)
    for i in range(len(log) - 1, -1, -1):
        x = log[i][1]
        term[i] = term[i + 1] + (flips[i] / x.numel() if (x is not None and flips[i]) else 0.0)
    out = {}
    for i, (names, _) in enumerate(log):
        for k in names:
            out[k] = 0.8 * term[i + 1] ** 0.5
    return out


def params_upstream_of(t: torch.Tensor):
    """ids of the leaf tensors (parameters) the autograd graph of ``t`` reaches."""
    seen, out, stack = set(), set(), [t.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        v = getattr(fn, "variable", None)
        if v is not None:
            out.add(id(v))
        for nf, _ in fn.next_functions:
            stack.append(nf)
    return out


def feature_map_flips(got_feats, ref_feats, model):
    """Discriminators: flips per parameter from the feature maps.  ``got_feats`` (HIP, still attached to their autograd
    graph) and ``ref_feats`` (the other evaluation) are lists (one per net) of lists of maps; the last map of a net is the
    score (its gate is the hinge, injected from one place by the tests), every other map is a LeakyReLU gate.
    Returns ({parameter name: flipped gate elements downstream of it}, total flips, total gate elements)."""
    names = {id(p): k for k, p in model.named_parameters()}
    down = {k: 0 for k in names.values()}
    total_flips = total = 0
    for net, rnet in zip(got_feats, ref_feats):
        for f, rf in zip(net[:-1], rnet[:-1]):
            d = (f.detach().cpu() > 0) != (rf.detach().cpu() > 0)
            n = int(d.sum())
            total += f.numel()
            if not n:
                continue
            total_flips += n
            for pid in params_upstream_of(f):
                if pid in names:
                    down[names[pid]] += n
    return down, total_flips, total


def feature_map_flip_effects(got_feats, ref_feats, model, grad_feats, factor: float = 0.9):
    """Discriminators: the first-order size of what the counted flips do to each parameter gradient.  ``grad_feats`` = the
    feature maps of an evaluation whose maps RETAINED their gradients (``f.retain_grad()`` before its backward pass; the tests
    pass the fp32 oracle's -- the HIP modules hand out re-laid-out views of their maps, which are not on the gradient's path):
    a flipped LeakyReLU gate on element e of map f multiplies the gradient that flows back through e by slope^(+-1), i.e.
    changes it by at most ``factor`` = 1 - slope (0.9 bounds the slopes 0.1 and 0.2 in use) of |dL/df[e]| -- relative to the norm
    of the whole gradient that leaves the map, ||dL/df * act'||.  With the sparse hinge cotangent of a discriminator step a single
    element can carry percents of that norm, which a numel-based estimate (0.9 / sqrt(numel)) misses by an order of magnitude.
    Returns {parameter name: sum over the flipped gates downstream of it of that relative change} (0 without a flip)."""
    names = {id(p): k for k, p in model.named_parameters()}
    out = {k: 0.0 for k in names.values()}
    for net, rnet, gnet in zip(got_feats, ref_feats, grad_feats):
        for f, rf, gf in zip(net[:-1], rnet[:-1], gnet[:-1]):
            d = (f.detach().cpu() > 0) != (rf.detach().cpu() > 0)
            if not bool(d.any()):
                continue
            assert gf.grad is not None, "feature_map_flip_effects: the maps of grad_feats did not retain their gradients"
            g = gf.grad.detach().cpu().double()
            slope = torch.where(gf.detach().cpu() > 0, torch.ones_like(g), torch.full_like(g, 1.0 - factor))
            through = float((g * slope).norm())
            eff = factor * float(g[d].abs().sum()) / max(through, 1e-300)
            for pid in params_upstream_of(f):
                if pid in names:
                    out[names[pid]] += eff
    return out
