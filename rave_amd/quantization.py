"""Drop-in mirror of rave/quantization.py (EuclideanCodebook, VectorQuantization,
ResidualVectorQuantization) on the HIP VQ kernels (rave_amd/csrc/vq.hip).  Same module tree and
buffers (``layers.N._codebook.{inited,cluster_size,embed,embed_avg}``) as the reference, so
checkpoints are interchangeable.

ResidualVectorQuantization.forward runs the whole residual chain in (N, D) row-major layout: one
transpose in, then per quantiser ONE assign launch (nearest code + residual + running sum + commit-loss
partials) and, in training, one EMA-update launch pair; one transpose out.

Reference quirks kept / noted:
  * the straight-through estimator makes d(quantized_out)/dz the identity and only the FIRST
    quantiser's commitment loss reaches z (residual_{i+1} = residual_i - (residual_i + const) has zero
    gradient w.r.t. residual_i) -- implemented as a custom backward, checked against autograd of the
    oracle restatement;
  * ``expire_codes_`` (quantization.py:107-115) overwrites rows of ``embed`` that the EMA update at the
    end of the same forward overwrites again (:176-179), so it has no effect on any result; it is skipped;
  * k-means initialisation (quantization.py:36-56,101-106) draws with torch.randperm on the device: it
    is reproduced with the HIP assign kernel + torch index_add, but is not bit-comparable to a CPU run.
    Parity tests load initialised codebooks.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from .ops import _chk


def _assign(x: torch.Tensor, embed: torch.Tensor, residual, qsum, want_loss: bool):
    n, d = x.shape
    k = embed.shape[0]
    ind = torch.empty(n, device=x.device, dtype=torch.int64)
    parts = torch.empty(L.lib.rh_vq_loss_partials(n), device=x.device, dtype=torch.float32) if want_loss else None
    nbytes = L.lib.rh_vq_assign_workspace_bytes(n, d, k)      # > 0: distance search on the f32 matrix cores (vq.hip, round 5)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32) if nbytes > 0 else None
    L.check(L.lib.rh_vq_assign_ws_f32(L.ptr(x), L.ptr(embed), n, d, k, L.ptr(ind), L.ptr(residual), L.ptr(qsum), L.ptr(parts),
                                      L.ptr(ws), nbytes, L.stream()), "vq_assign")
    return ind, parts


def sample_vectors(samples, num: int):
    n = samples.shape[0]
    if n >= num:
        idx = torch.randperm(n, device=samples.device)[:num]
    else:
        idx = torch.randint(0, n, (num,), device=samples.device)
    return samples[idx]


def kmeans(samples, num_clusters: int, num_iters: int = 10):
    """rave/quantization.py:36-56 with the (N, K, D) difference tensor replaced by the assign kernel."""
    samples = _chk(samples, "samples")
    means = sample_vectors(samples, num_clusters).contiguous()
    bins = torch.zeros(num_clusters, device=samples.device, dtype=torch.int64)
    for _ in range(num_iters):
        buckets, _ = _assign(samples, means, None, None, False)
        bins = torch.bincount(buckets, minlength=num_clusters)
        zero = bins == 0
        new = torch.zeros_like(means).index_add_(0, buckets, samples) / bins.clamp(min=1)[:, None]
        means = torch.where(zero[:, None], means, new).contiguous()
    return means, bins


class EuclideanCodebook(nn.Module):
    """rave/quantization.py:59-181."""

    def __init__(self, dim: int, codebook_size: int, kmeans_init: int = False, kmeans_iters: int = 10,
                 decay: float = 0.99, epsilon: float = 1e-5, threshold_ema_dead_code: int = 2):
        super().__init__()
        self.decay = decay
        if kmeans_init:
            embed = torch.zeros(codebook_size, dim)
        else:
            embed = torch.empty(codebook_size, dim)
            nn.init.kaiming_uniform_(embed)
        self.codebook_size = codebook_size
        self.kmeans_iters = kmeans_iters
        self.epsilon = epsilon
        self.threshold_ema_dead_code = threshold_ema_dead_code
        self.register_buffer("inited", torch.Tensor([not kmeans_init]))
        from .blocks import track_flag_buffers
        track_flag_buffers(self)
        self.register_buffer("cluster_size", torch.zeros(codebook_size))
        self.register_buffer("embed", embed)
        self.register_buffer("embed_avg", embed.clone())

    def init_embed_(self, data):
        embed, cluster_size = kmeans(data, self.codebook_size, self.kmeans_iters)
        self.embed.data.copy_(embed)
        self.embed_avg.data.copy_(embed)
        self.cluster_size.data.copy_(cluster_size)
        self.inited.data.copy_(torch.Tensor([True]))

    def step(self, x2d: torch.Tensor, residual, qsum, want_loss: bool):
        """One fused codebook step on (N, D) vectors; returns (indices, loss partials)."""
        from .blocks import host_flag
        if not host_flag(self, "inited"):
            self.init_embed_(x2d)
        ind, parts = _assign(x2d, self.embed, residual, qsum, want_loss)
        return ind, parts

    def ema_update_(self, x2d: torch.Tensor, ind: torch.Tensor):
        n, d = x2d.shape
        L.check(L.lib.rh_vq_ema_update_f32(L.ptr(x2d), L.ptr(ind), n, d, self.codebook_size, self.decay, self.epsilon,
                                           L.ptr(self.cluster_size), L.ptr(self.embed_avg), L.ptr(self.embed), L.stream()),
                "vq_ema_update")

    def encode(self, x):
        shape = x.shape
        ind, _ = _assign(_chk(x.reshape(-1, shape[-1]), "x"), self.embed, None, None, False)
        return ind.reshape(shape[0], shape[1])

    def decode(self, embed_ind):
        return nn.functional.embedding(embed_ind, self.embed)


class VectorQuantization(nn.Module):
    """rave/quantization.py:184-272 (codebook_dim == dim only: no projection, as every shipped config)."""

    def __init__(self, dim: int, codebook_size: int, codebook_dim: Optional[int] = None, decay: float = 0.99,
                 epsilon: float = 1e-5, kmeans_init: bool = True, kmeans_iters: int = 50,
                 threshold_ema_dead_code: int = 2, commitment_weight: float = 1.):
        super().__init__()
        if (codebook_dim or dim) != dim:
            raise NotImplementedError("rave_amd VectorQuantization: codebook_dim != dim (projection) is unused upstream")
        self.project_in = nn.Identity()
        self.project_out = nn.Identity()
        self.epsilon = epsilon
        self.commitment_weight = commitment_weight
        self._codebook = EuclideanCodebook(dim=dim, codebook_size=codebook_size, kmeans_init=kmeans_init,
                                           kmeans_iters=kmeans_iters, decay=decay, epsilon=epsilon,
                                           threshold_ema_dead_code=threshold_ema_dead_code)
        self.codebook_size = codebook_size

    @property
    def codebook(self):
        return self._codebook.embed

    def encode(self, x):
        return self._codebook.encode(x.permute(0, 2, 1))

    def decode(self, embed_ind):
        return self._codebook.decode(embed_ind).permute(0, 2, 1)

    def forward(self, x):
        out, loss, ind = _rvq_forward([self], x, self.training)
        return out, ind[:, 0], loss.reshape(1)


class _RvqFn(torch.autograd.Function):
    """Straight-through residual quantisation: forward values from the kernels, backward = identity on the
    quantised output + the first quantiser's commitment-loss gradient."""

    @staticmethod
    def forward(ctx, z, layers, training: bool):
        z = _chk(z, "z")
        b, d, t = z.shape
        x = z.permute(0, 2, 1).contiguous().reshape(b * t, d)       # (N, D)
        res = x.clone()
        qsum = torch.zeros_like(x)
        loss = torch.zeros((), device=z.device, dtype=torch.float32)
        inds = []
        first_res = None
        for i, layer in enumerate(layers):
            cb = layer._codebook
            cur = res if not training else res.clone()                # EMA update needs this quantiser's input
            ind, parts = cb.step(cur, res, qsum, training and layer.commitment_weight > 0)
            if i == 0 and training:
                first_res = res.clone()                               # x - q_0
            if training:
                cb.ema_update_(cur, ind)
                if parts is not None:
                    loss = loss + parts.sum() / cur.numel() * layer.commitment_weight
            inds.append(ind.reshape(b, t))
        out = qsum.reshape(b, t, d).permute(0, 2, 1).contiguous()
        ctx.training = training
        ctx.shape = (b, d, t)
        ctx.w0 = float(layers[0].commitment_weight) if training else 0.0
        ctx.save_for_backward(first_res if first_res is not None else torch.empty(0, device=z.device))
        ind_all = torch.stack(inds, 1)
        ctx.mark_non_differentiable(ind_all)
        return out, loss, ind_all

    @staticmethod
    def backward(ctx, dout, dloss, _dind):
        if not ctx.training:
            return None, None, None
        (first_res,) = ctx.saved_tensors
        b, d, t = ctx.shape
        dz = dout
        if ctx.w0 > 0 and dloss is not None:
            # d/dx mean((q0.detach() - x)^2) = 2 (x - q0) / numel
            g = (first_res * (2.0 * ctx.w0 / first_res.numel())).reshape(b, t, d).permute(0, 2, 1)
            dz = dz + g * dloss
        return dz, None, None


def _rvq_forward(layers, z, training: bool):
    if z.is_cuda:
        return _RvqFn.apply(z, list(layers), training)
    raise RuntimeError("rave_amd: RVQ input must live on the GPU (the HIP hot path has no CPU fallback)")


class ResidualVectorQuantization(nn.Module):
    """rave/quantization.py:275-322."""

    def __init__(self, num_quantizers, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([VectorQuantization(**kwargs) for _ in range(num_quantizers)])

    def forward(self, x):
        return _rvq_forward(self.layers, x, self.training)

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            _, _, ind = _rvq_forward(self.layers, x, False)
        return ind

    def decode(self, q_indices: torch.Tensor) -> torch.Tensor:
        out = torch.tensor(0.0, device=q_indices.device)
        for i, layer in enumerate(self.layers):
            out = out + layer.decode(q_indices[:, i])
        return out
