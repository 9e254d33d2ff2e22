"""Drop-in mirrors of rave/discriminator.py (ConvNet, MultiScale / MultiPeriod / Combine) on the
HIP kernels.  Feature lists are identical to the reference's: the conv outputs BEFORE LeakyReLU,
including the final 1x1 conv (rave/discriminator.py:113-119); the LeakyReLU(.2) between layers
is fused into the NEXT conv's input staging, MultiPeriodDiscriminator.fold's zero pad + reshape
and MultiScaleDiscriminator's avg_pool1d run in-kernel / as a HIP kernel.
"""
from __future__ import annotations

from typing import Sequence, Type

import numpy as np
import torch
import torch.nn as nn

from . import cc, ops
from .blocks import normalization
from .ops import ACT_LEAKY, ACT_NONE


def _map_conv(conv):
    """gin binds ``conv = @torch.nn.Conv1d`` / ``@nn.Conv2d`` (v1.gin:82-84, v2.gin:53-56).  What arrives is the class
    itself (plain Python), a gin-made subclass of it, or a configurable reference wrapping it -- all are mapped to the
    HIP operator classes (same constructor signature, same ``weight`` / ``bias`` parameter names)."""
    base = conv
    for attr in ("__gin_target__", "__wrapped__", "wrapped"):
        base = getattr(base, attr, base)
    if isinstance(base, type):
        if issubclass(base, (nn.Conv1d, cc.PlainConv1d)):
            return cc.PlainConv1d
        if issubclass(base, (nn.Conv2d, cc.Conv2dK1)):
            return cc.Conv2dK1
    # configurable references that only carry a name: an explicit allow-list (torch.nn and the cc classes) -- a
    # cached_conv-style class with other constructor / padding semantics must not be mapped silently
    name = getattr(base, "__name__", "") or getattr(base, "__qualname__", "")
    mod = getattr(base, "__module__", "") or ""
    known = mod.startswith(("torch.nn", "rave_amd.cc")) or mod in ("cached_conv", "cached_conv.convs")
    if known and name in ("Conv1d", "PlainConv1d"):
        return cc.PlainConv1d
    if known and name in ("Conv2d", "Conv2dK1"):
        return cc.Conv2dK1
    raise NotImplementedError(f"rave_amd ConvNet: conv class {conv} (expected torch.nn.Conv1d / torch.nn.Conv2d)")


class ConvNet(nn.Module):
    """rave/discriminator.py:77-119."""

    def __init__(self, in_size, out_size, capacity, n_layers, kernel_size, stride, conv) -> None:
        super().__init__()
        conv = _map_conv(conv)
        channels = [in_size] + list(capacity * 2 ** np.arange(n_layers))
        if isinstance(stride, int):
            stride = n_layers * [stride]
        net = []
        for i in range(n_layers):
            if not isinstance(kernel_size, int):
                pad = (cc.get_padding(kernel_size[0], stride[i], mode="centered")[0], 0)
                s = (stride[i], 1)
            else:
                pad = cc.get_padding(kernel_size, stride[i], mode="centered")[0]
                s = stride[i]
            net.append(normalization(conv(int(channels[i]), int(channels[i + 1]), kernel_size, stride=s, padding=pad)))
            net.append(nn.LeakyReLU(.2))
        net.append(conv(int(channels[-1]), out_size, 1))
        self.net = nn.Sequential(*net)

    def forward(self, x, period=None):
        if period is not None and _period_major():
            return self._forward_period_major(x, int(period))
        features = []
        act, slope = ACT_NONE, 0.0
        first = True
        for layer in self.net:
            if isinstance(layer, nn.LeakyReLU):
                act, slope = ACT_LEAKY, float(layer.negative_slope)   # fused into the next conv
                continue
            if first and period is not None:
                x = layer(x, act=act, slope=slope, period=period)
            else:
                x = layer(x, act=act, slope=slope)
            first = False
            act, slope = ACT_NONE, 0.0
            features.append(x)
        return features


    def _forward_period_major(self, x, period: int):
        """MultiPeriodDiscriminator.fold (rave/discriminator.py:192-195) with the period axis stored OUTERMOST:
        the (B, C, H, W) plane lives as (B * W, C, H), so that every (k,1) convolution of the stack is a plain
        strided Conv1d over contiguous rows -- the geometry the bf16x6 forward / data-gradient / weight-gradient
        kernels take (with W innermost the strided layers and every weight gradient stayed on the f32-MFMA kernels:
        25 of 82 ms of the v2 discriminator pass).  The feature maps handed back are (B, C, H, W) VIEWS of those
        tensors: same shapes and values as the reference's, no copy; elementwise losses keep the layout, so the
        gradients arrive period-major as well."""
        b, c, t = x.shape
        pad = (-t) % period
        if pad:
            x = nn.functional.pad(x, (0, pad))
        h = x.shape[-1] // period
        x = x.reshape(b, c, h, period).permute(0, 3, 1, 2).contiguous().view(b * period, c, h)
        features = []
        act, slope = ACT_NONE, 0.0
        for layer in self.net:
            if isinstance(layer, nn.LeakyReLU):
                act, slope = ACT_LEAKY, float(layer.negative_slope)   # fused into the next conv
                continue
            x = layer(x, act=act, slope=slope, period_major=True)
            act, slope = ACT_NONE, 0.0
            features.append(x.view(b, period, x.shape[1], x.shape[2]).permute(0, 2, 3, 1))
        return features


def _period_major() -> bool:
    import os
    return os.environ.get("RH_MPD_PERIOD_MAJOR", "1") != "0"


class MultiScaleDiscriminator(nn.Module):
    """rave/discriminator.py:122-136."""

    def __init__(self, n_discriminators, convnet, n_channels=1) -> None:
        super().__init__()
        self.layers = nn.ModuleList([convnet(in_size=n_channels) for _ in range(n_discriminators)])

    def forward(self, x):
        features = []
        for i, layer in enumerate(self.layers):
            features.append(layer(x))
            if i + 1 < len(self.layers):
                x = ops.avg_pool2(x)
        return features


class MultiPeriodDiscriminator(nn.Module):
    """rave/discriminator.py:174-195."""

    def __init__(self, periods, convnet, n_channels=1) -> None:
        super().__init__()
        self.periods = periods
        self.layers = nn.ModuleList([convnet(in_size=n_channels) for _ in periods])

    def forward(self, x):
        return [layer(x, period=n) for layer, n in zip(self.layers, self.periods)]


class CombineDiscriminators(nn.Module):
    """rave/discriminator.py:198-209."""

    def __init__(self, discriminators: Sequence[Type[nn.Module]], n_channels=1) -> None:
        super().__init__()
        self.discriminators = nn.ModuleList(d(n_channels=n_channels) for d in discriminators)

    def forward(self, x):
        features = []
        for disc in self.discriminators:
            features.extend(disc(x))
        return features


# --------------------------------------------------------------------------- spectral discriminator
class Spectrogram(nn.Module):
    """torchaudio.transforms.Spectrogram(power=None) as the reference configures it; owns the same
    ``window`` buffer (state_dict key ``...window``).  Framing + FFT stay on torch.stft / rocFFT."""

    def __init__(self, n_fft: int, hop_length: int, normalized: bool, center: bool) -> None:
        super().__init__()
        self.n_fft, self.hop_length, self.normalized, self.center = n_fft, hop_length, normalized, center
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        s = torch.stft(x.reshape(-1, shape[-1]), self.n_fft, hop_length=self.hop_length, win_length=self.n_fft,
                       window=self.window, center=self.center, pad_mode="reflect", normalized=False, onesided=True,
                       return_complex=True)
        if self.normalized:   # torchaudio "window" normalisation
            s = s / self.window.pow(2.).sum().sqrt()
        return s.reshape(shape[:-1] + s.shape[-2:])


def spectrogram(n_fft: int) -> Spectrogram:
    """rave/discriminator.py:12-20."""
    return Spectrogram(n_fft, hop_length=n_fft // 4, normalized=True, center=False)


def rectified_2d_conv_block(capacity, kernel_sizes, strides=None, dilations=None, in_size=None, out_size=None,
                            activation: bool = True):
    """rave/discriminator.py:23-51 (same module tree / state_dict keys) on cc.Conv2d."""
    if dilations is None:
        paddings = kernel_sizes[0] // 2, kernel_sizes[1] // 2
    else:
        fks = (kernel_sizes[0] - 1) * dilations[0], (kernel_sizes[1] - 1) * dilations[1]
        paddings = fks[0] // 2, fks[1] // 2
    conv = normalization(cc.Conv2d(in_size or capacity, out_size or capacity, kernel_size=kernel_sizes,
                                   stride=strides or (1, 1), dilation=dilations or (1, 1), padding=paddings))
    if not activation:
        return conv
    return nn.Sequential(conv, nn.LeakyReLU(.2))


def run_conv2d_layer(layer, x):
    """conv (+ LeakyReLU fused into the conv's epilogue) of a WNConv2d / rectified block."""
    if isinstance(layer, nn.Sequential):
        return layer[0](x, act=ACT_LEAKY, slope=float(layer[1].negative_slope))
    return layer(x)


# The Encodec spectral net IS its layer table -- kernel, stride, dilation per block (rave/discriminator.py:57-66; the drop-in contract
# is "same module tree, same state_dict keys", so the table is necessarily the reference's): (9,3) stem on the (real, imag)
# planes, three (9,3) blocks striding 2 along frequency with time dilation 1 / 2 / 4, a (3,3) block and the (3,3) scoring conv.
ENCODEC_BLOCKS = (
    # kernel, stride,  dilation, role
    ((9, 3), None,     None,     "stem"),
    ((9, 3), (2, 1),   (1, 1),   "body"),
    ((9, 3), (2, 1),   (1, 2),   "body"),
    ((9, 3), (2, 1),   (1, 4),   "body"),
    ((3, 3), None,     None,     "body"),
    ((3, 3), None,     None,     "score"),
)


class EncodecConvNet(nn.Module):
    """rave/discriminator.py:54-74: the blocks of ENCODEC_BLOCKS in ``net`` (indices = the reference's state_dict keys); forward
    returns every block's output (post-activation for all but the scoring conv), each block one HIP conv launch with its
    LeakyReLU in the epilogue (run_conv2d_layer)."""

    def __init__(self, capacity: int, n_channels: int = 1) -> None:
        super().__init__()
        blocks = []
        for kernel, stride, dilation, role in ENCODEC_BLOCKS:
            blocks.append(rectified_2d_conv_block(capacity, kernel, stride, dilation,
                                                  in_size=2 * n_channels if role == "stem" else None,
                                                  out_size=1 if role == "score" else None, activation=role != "score"))
        self.net = nn.Sequential(*blocks)

    def forward(self, x):
        features = []
        for layer in self.net:
            x = run_conv2d_layer(layer, x)
            features.append(x)
        return features


class MultiScaleSpectralDiscriminator(nn.Module):
    """rave/discriminator.py:139-153."""

    def __init__(self, scales: Sequence[int], convnet, n_channels: int = 1) -> None:
        super().__init__()
        self.specs = nn.ModuleList([spectrogram(n) for n in scales])
        self.nets = nn.ModuleList([convnet(n_channels=n_channels) for _ in scales])

    def forward(self, x):
        features = []
        for spec, net in zip(self.specs, self.nets):
            spec_x = spec(x)
            spec_x = torch.cat([spec_x.real, spec_x.imag], 1)
            features.append(net(spec_x))
        return features
