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
from .discriminator import _period_major, run_conv2d_layer

# frequency bands of the multi-resolution nets as fractions of the spectrum (descript_discriminator.py:115)
BANDS = [(0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)]

_LEAK = 0.1
# (out channels, stride along the folded time axis) of the period nets' (5,1) convolutions (:36-42)
_PERIOD_STACK = ((32, 3), (128, 3), (512, 3), (1024, 3), (1024, 1))
# (kernel, stride) of the band nets' convolutions, all 32 channels wide, "same" zero padding (:137-143)
_BAND_STACK = (((3, 9), (1, 1)), ((3, 9), (1, 2)), ((3, 9), (1, 2)), ((3, 9), (1, 2)), ((3, 3), (1, 1)))
_BAND_WIDTH = 32


def WNConv2d(*args, **kwargs):
    """rave/descript_discriminator.py:22-27: weight-normalised Conv2d, followed by LeakyReLU(0.1) unless act=False.
    Same module tree as upstream (Sequential(conv, LeakyReLU) / bare conv) so that state_dict keys agree; the
    activation is fused into the conv's epilogue at run time (run_conv2d_layer)."""
    with_act = kwargs.pop("act", True)
    conv = weight_norm(cc.Conv2d(*args, **kwargs))
    return nn.Sequential(conv, nn.LeakyReLU(_LEAK)) if with_act else conv


class MPD(nn.Module):
    """rave/descript_discriminator.py:30-66: reflect-pad to a multiple of the period, fold to (B, C, T/p, p),
    five (5,1) convolutions + a (3,1) scoring conv; returns every (activated) feature map."""

    def __init__(self, period, n_channels: int = 1):
        super().__init__()
        self.period = period
        widths = (n_channels,) + tuple(c for c, _ in _PERIOD_STACK)
        self.convs = nn.ModuleList(
            WNConv2d(widths[i], widths[i + 1], (5, 1), (stride, 1), padding=(2, 0))
            for i, (_, stride) in enumerate(_PERIOD_STACK))
        self.conv_post = WNConv2d(widths[-1], 1, kernel_size=(3, 1), padding=(1, 0), act=False)

    def pad_to_period(self, x):
        # upstream pads by period - t % period, i.e. a FULL period when t is already a multiple (:50-53)
        return F.pad(x, (0, self.period - x.shape[-1] % self.period), mode="reflect")

    def forward(self, x):
        x = self.pad_to_period(x)
        b, c, t = x.shape
        if _period_major():
            # the period axis stored OUTERMOST: (B, C, H, W) lives as (B * W, C, H), every (5,1) convolution is a plain
            # strided Conv1d over contiguous rows (the geometry the bf16x6 data- and weight-gradient kernels take; with W
            # innermost the weight gradients of this stack -- 42 of 175 ms -- stayed on the f32-MFMA kernel).  The
            # feature maps handed back are (B, C, H, W) views: same shapes and values as the reference's, no copy.
            p = self.period
            x = x.reshape(b, c, t // p, p).permute(0, 3, 1, 2).contiguous().view(b * p, c, t // p)
            fmap = []
            for layer in list(self.convs) + [self.conv_post]:
                if isinstance(layer, nn.Sequential):
                    x = layer[0](x, act=ops.ACT_LEAKY, slope=float(layer[1].negative_slope), period_major=True)
                else:
                    x = layer(x, period_major=True)
                fmap.append(x.view(b, p, x.shape[1], x.shape[2]).permute(0, 2, 3, 1))
            return fmap
        x = x.reshape(b, c, t // self.period, self.period)
        fmap = []
        for layer in list(self.convs) + [self.conv_post]:
            x = run_conv2d_layer(layer, x)
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
    """rave/descript_discriminator.py:118-184: complex STFT as (real, imag) channels in (frames, bins) layout, one
    5-conv net per frequency band, band outputs concatenated along frequency into a (3,3) scoring conv."""

    def __init__(self, window_length: int, hop_factor: float = 0.25, sample_rate: int = 44100, bands: list = BANDS,
                 n_channels: int = 1):
        super().__init__()
        self.window_length, self.hop_factor, self.sample_rate = window_length, hop_factor, sample_rate
        n_bins = window_length // 2 + 1
        self.bands = [(int(lo * n_bins), int(hi * n_bins)) for lo, hi in bands]

        def band_net():
            widths = (2 * n_channels,) + (_BAND_WIDTH,) * len(_BAND_STACK)
            return nn.ModuleList(
                WNConv2d(widths[i], widths[i + 1], k, st, padding=(k[0] // 2, k[1] // 2))
                for i, (k, st) in enumerate(_BAND_STACK))

        self.band_convs = nn.ModuleList(band_net() for _ in self.bands)
        self.conv_post = WNConv2d(_BAND_WIDTH, 1, (3, 3), (1, 1), padding=(1, 1), act=False)
        self.stft = _Stft(window_length, int(hop_factor * window_length))

    def spectrogram(self, x):
        s = torch.view_as_real(self.stft(x))                      # (b, c, frames, bins, re/im)
        b, c, t, f, p = s.shape
        s = s.permute(0, 1, 4, 2, 3).reshape(b, c * p, t, f)      # upstream: "b c f t p -> b (c p) t f"
        return [s[..., lo:hi] for lo, hi in self.bands]

    def forward(self, x):
        fmap, tails = [], []
        for band, net in zip(self.spectrogram(x), self.band_convs):
            for layer in net:
                band = run_conv2d_layer(layer, band)
                fmap.append(band)
            tails.append(band)
        fmap.append(run_conv2d_layer(self.conv_post, torch.cat(tails, dim=-1)))
        return fmap


class DescriptDiscriminator(nn.Module):
    """rave/descript_discriminator.py:187-217 (configs/descript_discriminator.gin): periods x MPD + fft sizes x MRD
    on the DC-free, peak-normalised waveform."""

    def __init__(self, rates: list = [], periods: list = [2, 3, 5, 7, 11], fft_sizes: list = [2048, 1024, 512],
                 sample_rate: int = 44100, bands: list = BANDS, n_channels: int = 1):
        super().__init__()
        if rates:
            raise NotImplementedError("rave_amd DescriptDiscriminator: MSD (rates != []) is unreachable upstream too")
        nets = [MPD(p, n_channels=n_channels) for p in periods]
        nets.extend(MRD(f, sample_rate=sample_rate, bands=bands, n_channels=n_channels) for f in fft_sizes)
        self.discriminators = nn.ModuleList(nets)

    def preprocess(self, y):
        centred = y - y.mean(dim=-1, keepdim=True)
        peak = centred.abs().max(dim=-1, keepdim=True)[0]
        return 0.8 * centred / (peak + 1e-9)

    def forward(self, x):
        x = self.preprocess(x)
        return [net(x) for net in self.discriminators]
