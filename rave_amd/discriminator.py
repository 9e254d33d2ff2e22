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
    """gin binds ``conv = @torch.nn.Conv1d`` / ``@nn.Conv2d`` (v1.gin:82-84, v2.gin:53-56)."""
    if conv in (nn.Conv1d, cc.PlainConv1d):
        return cc.PlainConv1d
    if conv in (nn.Conv2d, cc.Conv2dK1):
        return cc.Conv2dK1
    raise NotImplementedError(f"rave_amd ConvNet: conv class {conv}")


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
