"""Drop-in mirror of rave/descript_discriminator.py (MPD, MRD, DescriptDiscriminator) with every
Conv2d on the HIP 2-D kernels (rave_amd/csrc/conv2d.hip) and the LeakyReLU(0.1) that follows each
conv fused into its epilogue.  Module tree and state_dict keys equal the reference's
(``discriminators.N.convs.M.0.weight_g`` ...).  The STFT of MRD uses the HIP framing kernel + rocFFT;
``preprocess`` (DC removal / peak normalisation, :206-211) and the reflect pad of MPD are PyTorch glue.

Not provided: ``MSD`` (:69-112) -- DescriptDiscriminator is only ever built with ``rates=[]``
(v3.gin), and the reference's own call passes a ``sample_rate`` keyword MSD.__init__ does not accept
(:201), so that branch cannot run upstream either.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cc, ops
from .blocks import weight_norm
from .discriminator import run_conv2d_layer

BANDS = [(0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)]


def WNConv2d(*args, **kwargs):
    """rave/descript_discriminator.py:22-27."""
    act = kwargs.pop("act", True)
    conv = weight_norm(cc.Conv2d(*args, **kwargs))
    if not act:
        return conv
    return nn.Sequential(conv, nn.LeakyReLU(0.1))


class MPD(nn.Module):
    """rave/descript_discriminator.py:30-66."""

    def __init__(self, period, n_channels: int = 1):
        super().__init__()
        self.period = period
        self.convs = nn.ModuleList([
            WNConv2d(n_channels, 32, (5, 1), (3, 1), padding=(2, 0)),
            WNConv2d(32, 128, (5, 1), (3, 1), padding=(2, 0)),
            WNConv2d(128, 512, (5, 1), (3, 1), padding=(2, 0)),
            WNConv2d(512, 1024, (5, 1), (3, 1), padding=(2, 0)),
            WNConv2d(1024, 1024, (5, 1), 1, padding=(2, 0)),
        ])
        self.conv_post = WNConv2d(1024, 1, kernel_size=(3, 1), padding=(1, 0), act=False)

    def pad_to_period(self, x):
        t = x.shape[-1]
        return F.pad(x, (0, self.period - t % self.period), mode="reflect")

    def forward(self, x):
        fmap = []
        x = self.pad_to_period(x)
        b, c, t = x.shape
        x = x.reshape(b, c, t // self.period, self.period)
        for layer in self.convs:
            x = run_conv2d_layer(layer, x)
            fmap.append(x)
        x = run_conv2d_layer(self.conv_post, x)
        fmap.append(x)
        return fmap


class _Stft(nn.Module):
    """torchaudio Spectrogram(n_fft=w, hop=w/4, center=True, power=None) -> (B, C, frames, bins) complex:
    HIP framing kernel (centre + reflect pad + Hann) + rocFFT.  Owns the ``window`` buffer."""

    def __init__(self, n_fft: int, hop: int) -> None:
        super().__init__()
        self.n_fft, self.hop = n_fft, hop
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, x):
        b, c, t = x.shape
        frames = ops.stft_frames(x.reshape(b * c, t), self.window, self.n_fft, self.hop)
        s = torch.fft.rfft(frames, dim=-1)
        return s.reshape(b, c, s.shape[-2], s.shape[-1])


class MRD(nn.Module):
    """rave/descript_discriminator.py:118-184."""

    def __init__(self, window_length: int, hop_factor: float = 0.25, sample_rate: int = 44100, bands: list = BANDS,
                 n_channels: int = 1):
        super().__init__()
        self.window_length = window_length
        self.hop_factor = hop_factor
        self.sample_rate = sample_rate
        n_fft = window_length // 2 + 1
        self.bands = [(int(b[0] * n_fft), int(b[1] * n_fft)) for b in bands]
        ch = 32
        convs = lambda: nn.ModuleList([   # noqa: E731
            WNConv2d(2 * n_channels, ch, (3, 9), (1, 1), padding=(1, 4)),
            WNConv2d(ch, ch, (3, 9), (1, 2), padding=(1, 4)),
            WNConv2d(ch, ch, (3, 9), (1, 2), padding=(1, 4)),
            WNConv2d(ch, ch, (3, 9), (1, 2), padding=(1, 4)),
            WNConv2d(ch, ch, (3, 3), (1, 1), padding=(1, 1)),
        ])
        self.band_convs = nn.ModuleList([convs() for _ in range(len(self.bands))])
        self.conv_post = WNConv2d(ch, 1, (3, 3), (1, 1), padding=(1, 1), act=False)
        self.stft = _Stft(window_length, int(hop_factor * window_length))

    def spectrogram(self, x):
        s = torch.view_as_real(self.stft(x))                      # (b, c, t, f, p)
        b, c, t, f, p = s.shape
        s = s.permute(0, 1, 4, 2, 3).reshape(b, c * p, t, f)      # "b c f t p -> b (c p) t f"
        return [s[..., lo:hi] for lo, hi in self.bands]

    def forward(self, x):
        x_bands = self.spectrogram(x)
        fmap = []
        x = []
        for band, stack in zip(x_bands, self.band_convs):
            for layer in stack:
                band = run_conv2d_layer(layer, band)
                fmap.append(band)
            x.append(band)
        x = torch.cat(x, dim=-1)
        x = run_conv2d_layer(self.conv_post, x)
        fmap.append(x)
        return fmap


class DescriptDiscriminator(nn.Module):
    """rave/descript_discriminator.py:187-217."""

    def __init__(self, rates: list = [], periods: list = [2, 3, 5, 7, 11], fft_sizes: list = [2048, 1024, 512],
                 sample_rate: int = 44100, bands: list = BANDS, n_channels: int = 1):
        super().__init__()
        if rates:
            raise NotImplementedError("rave_amd DescriptDiscriminator: MSD (rates != []) is unreachable upstream too")
        discs = [MPD(p, n_channels=n_channels) for p in periods]
        discs += [MRD(f, sample_rate=sample_rate, bands=bands, n_channels=n_channels) for f in fft_sizes]
        self.discriminators = nn.ModuleList(discs)

    def preprocess(self, y):
        y = y - y.mean(dim=-1, keepdims=True)
        y = 0.8 * y / (y.abs().max(dim=-1, keepdim=True)[0] + 1e-9)
        return y

    def forward(self, x):
        x = self.preprocess(x)
        return [d(x) for d in self.discriminators]
