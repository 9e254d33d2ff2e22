"""Drop-in for the ``cached_conv`` operator API the reference's blocks are written against
(third-party ``cached-conv>=2.5.0``, reference requirements.txt:14; call sites listed in
SURVEY.md section 2.2).  Same names, constructor arguments, attributes (``_pad``,
``cumulative_delay``) and ``state_dict`` layout; the arithmetic runs in librave_hip.so.

Only the non-streaming mode is provided (the only mode used in training; streaming is switched
on by scripts/export.py:493 for CPU export, which is out of scope): ``use_cached_conv(True)``
raises.

Use with the untouched reference:  ``import rave_amd.cc as cc_hip; sys.modules['cached_conv'] =
cc_hip`` before ``import rave`` -- blocks resolve ``cc.Conv1d`` at call time
(rave/blocks.py:64,96,...).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_SNAKE, ConvGeom  # noqa: F401

MAX_BATCH_SIZE = 64
USE_BUFFER_CONV = False


def use_cached_conv(state: bool) -> None:
    if state:
        raise NotImplementedError("rave_amd.cc: streaming (cached) convolutions are not part of the "
                                  "training hot path and are not implemented")


_DEFAULT_PADDING_MODE = "centered"


def set_default_padding_mode(mode: str) -> None:
    """What ``cc.get_padding.mode = 'causal'`` (configs/causal.gin:5) does through gin: it changes
    the DEFAULT of ``mode``; explicit ``mode=`` arguments (rave/discriminator.py:91-97) win."""
    global _DEFAULT_PADDING_MODE
    if mode not in ("centered", "causal"):
        raise ValueError(mode)
    _DEFAULT_PADDING_MODE = mode


def get_padding(kernel_size: int, stride: int = 1, dilation: int = 1, mode: Optional[str] = None) -> Tuple[int, int]:
    """cc.get_padding -- 'same' padding; ``stride`` is ignored exactly as upstream
    (``causal.gin:5`` rebinds the default ``mode``)."""
    if mode is None:
        mode = _DEFAULT_PADDING_MODE
    if kernel_size == 1:
        return (0, 0)
    p = (kernel_size - 1) * dilation + 1
    if mode == "centered":
        return ((p - 1) // 2, p // 2)
    if mode == "causal":
        return (p // 2 + (p - 1) // 2, 0)
    raise Exception(f"Padding mode {mode} is not valid")


def _effective_weight(mod: nn.Module) -> torch.Tensor:
    """Materialised weight of a (possibly weight-normalised) conv (rh_weight_norm_fwd_f32)."""
    g = getattr(mod, "weight_g", None)
    if g is not None:
        return ops.weight_norm(mod.weight_v, g)
    return mod.weight


def _wn_pair(mod: nn.Module):
    """(weight, gain) as the fused ops take them: if ``weight_g``/``weight_v`` exist (registered by
    rave_amd.blocks.normalization or by torch.nn.utils.weight_norm) returns (v, g) and the product
    g v/||v|| is folded into the weight repack once per forward, like the reference recomputes it
    once per forward (rave/blocks.py:15-22); otherwise (weight, None)."""
    g = getattr(mod, "weight_g", None)
    if g is not None:
        return mod.weight_v, g
    return mod.weight, None


class Conv1d(nn.Conv1d):
    """cc.Conv1d: ``padding=(left, right)`` asymmetric zero padding done inside the kernel."""

    def __init__(self, *args, **kwargs):
        pad = kwargs.get("padding", (0, 0))
        if isinstance(pad, int):
            pad = (pad, pad)
        self._pad = tuple(int(p) for p in pad)
        kwargs.pop("cumulative_delay", 0)
        kwargs["padding"] = 0
        super().__init__(*args, **kwargs)
        self.cumulative_delay = 0

    def script_cache(self):
        pass

    def geom(self, act: int = ACT_NONE, slope: float = 0.2) -> ConvGeom:
        return ConvGeom(stride=self.stride[0], dilation=self.dilation[0], pad_left=self._pad[0],
                        pad_right=self._pad[1], act=act, slope=slope)

    def forward(self, x, act: int = ACT_NONE, slope: float = 0.2, alpha: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None):
        if self.groups != 1:
            # grouped conv (only the v1 Encoder head, rave/blocks.py:490-497, groups = n_out = 2 on a
            # (B, C, 32) latent): one kernel launch per group on channel slices
            if alpha is not None or residual is not None:
                raise NotImplementedError("rave_amd.cc.Conv1d: groups with fused snake/residual")
            w = _effective_weight(self)
            xs = x.chunk(self.groups, 1)
            ws = w.chunk(self.groups, 0)
            bs = self.bias.chunk(self.groups, 0) if self.bias is not None else [None] * self.groups
            return torch.cat([ops.conv1d(xi.contiguous(), wi.contiguous(), bi, geom=self.geom(act, slope))
                              for xi, wi, bi in zip(xs, ws, bs)], 1)
        w, g = _wn_pair(self)
        return ops.conv1d(x, w, self.bias, geom=self.geom(act, slope), alpha=alpha, residual=residual, weight_g=g,
                          prepacked=getattr(self, "_prepacked", None))


class ConvTranspose1d(nn.ConvTranspose1d):
    """cc.ConvTranspose1d (symmetric ``padding``), sub-pixel form in the kernel."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("cumulative_delay", 0)
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.dilation[0] != 1 or self.output_padding[0] != 0:
            raise NotImplementedError("rave_amd.cc.ConvTranspose1d: groups/dilation/output_padding")
        self.cumulative_delay = 0

    def script_cache(self):
        pass

    def geom(self, act: int = ACT_NONE, slope: float = 0.2) -> ConvGeom:
        return ConvGeom(stride=self.stride[0], pad_left=self.padding[0], pad_right=self.padding[0],
                        transposed=True, act=act, slope=slope)

    def forward(self, x, act: int = ACT_NONE, slope: float = 0.2, alpha: Optional[torch.Tensor] = None):
        w, g = _wn_pair(self)
        return ops.conv1d(x, w, self.bias, geom=self.geom(act, slope), alpha=alpha, weight_g=g,
                          prepacked=getattr(self, "_prepacked", None))


class PlainConv1d(nn.Conv1d):
    """torch.nn.Conv1d signature (symmetric int padding, bias=True) on the HIP kernels; what
    discriminator.ConvNet gets when gin binds ``conv = @torch.nn.Conv1d`` (v1.gin:82-84)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise NotImplementedError("rave_amd.cc.PlainConv1d: groups / padding_mode")

    def forward(self, x, act: int = ACT_NONE, slope: float = 0.2):
        g = ConvGeom(stride=self.stride[0], dilation=self.dilation[0], pad_left=self.padding[0],
                     pad_right=self.padding[0], act=act, slope=slope)
        w, wg = _wn_pair(self)
        return ops.conv1d(x, w, self.bias, geom=g, weight_g=wg, prepacked=getattr(self, "_prepacked", None))


class Conv2dK1(nn.Conv2d):
    """torch.nn.Conv2d restricted to kernels (k,1), stride (s,1), padding (p,0): the
    MultiPeriodDiscriminator convs (rave/discriminator.py:86-100, v2.gin:53-59).  Runs as a 1-D
    convolution over H with W contiguous 'inner' columns -- no transpose, no F.pad copy."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if (self.kernel_size[1] != 1 or self.stride[1] != 1 or self.padding[1] != 0 or self.groups != 1
                or self.dilation != (1, 1) or self.padding_mode != "zeros"):
            raise NotImplementedError("rave_amd.cc.Conv2dK1 supports kernel (k,1), stride (s,1), padding (p,0)")

    def forward(self, x, act: int = ACT_NONE, slope: float = 0.2, period: Optional[int] = None,
                period_major: bool = False):
        """``period`` set: x is the UNFOLDED (B, C, T) waveform; MultiPeriodDiscriminator.fold
        (zero pad to a multiple of the period + reshape, rave/discriminator.py:192-195) happens
        inside the kernel.  ``period_major``: x is (B * W, C, H) -- the (B, C, H, W) plane stored with the period
        axis outermost -- and the (k,1) convolution is a plain 1-D one over H (discriminator.ConvNet)."""
        if period_major:
            g = ConvGeom(stride=self.stride[0], dilation=1, pad_left=self.padding[0], pad_right=self.padding[0],
                         act=act, slope=slope)
            w, wg = _wn_pair(self)
            return ops.conv1d(x, w, self.bias, geom=g, weight_g=wg, prepacked=getattr(self, "_prepacked", None))
        inner = period if period is not None else x.shape[3]
        g = ConvGeom(stride=self.stride[0], dilation=1, pad_left=self.padding[0], pad_right=self.padding[0],
                     act=act, slope=slope, inner=inner, fold=period is not None)
        w, wg = _wn_pair(self)
        return ops.conv1d(x, w, self.bias, geom=g, weight_g=wg, prepacked=getattr(self, "_prepacked", None))


class Conv2d(nn.Conv2d):
    """torch.nn.Conv2d (zero padding, groups == 1, any kernel / stride / dilation) on the HIP 2-D kernels
    (rh_conv2d_*_f32).  ``act`` is applied to the OUTPUT: the spectral / descript discriminators keep
    LeakyReLU(conv(x)) as a feature map (rave/discriminator.py:50, rave/descript_discriminator.py:27)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise NotImplementedError("rave_amd.cc.Conv2d: groups / padding_mode")

    def forward(self, x, act: int = ACT_NONE, slope: float = 0.2, period_major: bool = False):
        if period_major:
            # x is (B * W, C, H): the (B, C, H, W) plane with the period axis outermost; a (k,1) convolution is then a
            # plain strided Conv1d over contiguous rows (descript MPD; see discriminator.ConvNet._forward_period_major)
            if not (self.kernel_size[1] == 1 and self.stride[1] == 1 and self.padding[1] == 0 and self.dilation == (1, 1)):
                raise NotImplementedError("rave_amd.cc.Conv2d: period-major layout needs a (k,1) convolution")
            g = ConvGeom(stride=self.stride[0], dilation=1, pad_left=self.padding[0], pad_right=self.padding[0],
                         out_act=act, out_slope=slope)
            w, wg = _wn_pair(self)
            return ops.conv1d(x, w, self.bias, geom=g, weight_g=wg)
        if (self.kernel_size[1] == 1 and self.stride[1] == 1 and self.padding[1] == 0 and self.dilation == (1, 1)
                and self.stride[0] <= 8):
            # (k,1) kernels (descript MPD: up to 1024 channels on a period-wide plane): the 1-D LDS-DMA
            # kernels with W as contiguous "inner" columns are ~4x faster than the general 2-D tiling there
            g = ConvGeom(stride=self.stride[0], dilation=1, pad_left=self.padding[0], pad_right=self.padding[0],
                         inner=x.shape[3], out_act=act, out_slope=slope)
            w, wg = _wn_pair(self)
            return ops.conv1d(x, w, self.bias, geom=g, weight_g=wg)
        return ops.conv2d(x, _effective_weight(self), self.bias, self.stride, self.padding, self.dilation, act, slope)


class CachedSequential(nn.Sequential):
    def __init__(self, *args, **kwargs):
        cumulative_delay = kwargs.pop("cumulative_delay", 0)
        stride = kwargs.pop("stride", 1)
        super().__init__(*args, **kwargs)
        for m in reversed(list(args)):
            if hasattr(m, "cumulative_delay"):
                cumulative_delay = m.cumulative_delay
                break
        self.cumulative_delay = cumulative_delay
        self.stride = stride


class AlignBranches(nn.Module):
    def __init__(self, *branches, delays=None, cumulative_delay=0, stride=1):
        super().__init__()
        self.branches = nn.ModuleList(branches)
        self.cumulative_delay = cumulative_delay

    def forward(self, x):
        return [branch(x) for branch in self.branches]


class CachedPadding1d(nn.Module):  # streaming-only upstream; kept for isinstance checks
    def __init__(self, padding, crop=False):
        super().__init__()
        self.padding = padding
        self.crop = crop


class convs:  # ``cc.convs.Conv1d`` is read by scripts/export_onnx.py:34
    Conv1d = Conv1d
    ConvTranspose1d = ConvTranspose1d
