"""Losses of RAVE.training_step (rave/core.py) -- beside the HIP hot path ("next" #1).

On the GPU one STFT scale is: HIP framing kernel (centre + reflect pad + Hann) -> rocFFT R2C called through
hipFFT directly (rave_amd/fft.py) -> ONE fused HIP kernel for magnitude, linear + log distance and the three
reductions; backward: fused gradient kernel emitting the operand of the C2R adjoint of rfft -> rocFFT C2R ->
HIP framing adjoint (rave_amd.ops.stft_distance).  Only the FFT itself is library code.
There is no CPU branch: the reference's torch formulation lives in oracle/rave_oracle.py (audio_distance_v1) and is what the
parity tests compare with.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn


def mean_difference(target, value, norm: str = "L1", relative: bool = False):
    """rave/core.py:236-252."""
    diff = target - value
    if norm == "L1":
        diff = diff.abs().mean()
        if relative:
            diff = diff / target.abs().mean()
        return diff
    if norm == "L2":
        diff = (diff * diff).mean()
        if relative:
            diff = diff / (target * target).mean()
        return diff
    raise Exception(f"Norm must be either L1 or L2, got {norm}")


def hinge_gan(score_real, score_fake):
    """rave/core.py:151-155."""
    loss_dis = (torch.relu(1 - score_real) + torch.relu(1 + score_fake)).mean()
    return loss_dis, -score_fake.mean()


class MultiScaleSTFT(nn.Module):
    """rave/core.py:269-319 with magnitude=True, no mel; torchaudio.transforms.Spectrogram(n_fft=s,
    win_length=s, hop_length=s//4, power=None) == torch.stft(periodic Hann, center, reflect)."""

    def __init__(self, scales: Sequence[int], sample_rate: int = 44100, magnitude: bool = True):
        super().__init__()
        if not magnitude:
            raise NotImplementedError
        self.scales = list(scales)
        for s in self.scales:
            self.register_buffer(f"window_{s}", torch.hann_window(s), persistent=False)

    def complex_stfts(self, x):
        """Complex spectrograms (rows, frames, bins): torch.stft's with the (frequency, frame) axes swapped."""
        x = x.reshape(-1, x.shape[-1])
        # framing (centre + reflect pad + Hann window) in one HIP pass (raises on a CPU tensor), FFT on rocFFT
        from . import ops
        return [torch.fft.rfft(ops.stft_frames(x, getattr(self, f"window_{s}"), s, s // 4), dim=-1) for s in self.scales]

    def forward(self, x):
        """Magnitudes in torch.stft's (rows, bins, frames) layout."""
        return [y.abs().transpose(-1, -2) for y in self.complex_stfts(x)]


class AudioDistanceV1(nn.Module):
    """rave/core.py:322-344."""

    def __init__(self, multiscale_stft, log_epsilon: float) -> None:
        super().__init__()
        self.multiscale_stft = multiscale_stft()
        self.log_epsilon = log_epsilon

    def forward(self, x, y):
        if not (x.is_cuda and y.is_cuda):
            raise RuntimeError("rave_amd AudioDistanceV1: inputs must live on the GPU (the HIP path has no CPU fallback)")
        # STFT, magnitudes, both distances and their reductions inside one HIP kernel per scale (rh_stft_loss_*_f32)
        from . import ops
        ms = self.multiscale_stft
        xr, yr = x.reshape(-1, x.shape[-1]), y.reshape(-1, y.shape[-1])
        # all scales in ONE autograd node: the scale sum and the per-signal gradient accumulation happen inside
        distance = ops.multiscale_stft_distance(xr, yr, [getattr(ms, f"window_{s}") for s in ms.scales], ms.scales,
                                                float(self.log_epsilon))
        return {"spectral_distance": distance}


# ---- the other distances of rave/core.py (:415-490); no shipped config selects them (v1 ... v3, discrete use AudioDistanceV1)
class WaveformDistance(nn.Module):
    """rave/core.py:433-440."""

    def __init__(self, norm: str) -> None:
        super().__init__()
        self.norm = norm

    def forward(self, x, y):
        return mean_difference(y, x, self.norm)


class SpectralDistance(nn.Module):
    """rave/core.py:443-490: torchaudio Spectrogram(n_fft, hop = n_fft / 4, power, normalized, center=False) of both signals,
    then ``mean_difference(y, x, norm)`` summed over the norms.  Beside the hot path and unused by the shipped configs: framing
    is a strided view, the transform rocFFT (torch.fft), the reductions ATen -- library code on whatever device the signals
    live on; nothing here is a HIP kernel of this package.  ``mel`` (torchaudio MelSpectrogram, hybrid.gin) is not built."""

    def __init__(self, n_fft: int, sampling_rate: int, norm, power, normalized: bool, mel=None) -> None:
        super().__init__()
        if mel:
            raise NotImplementedError("rave_amd SpectralDistance: the mel variant (torchaudio MelSpectrogram) is not built")
        self.n_fft, self.hop, self.power, self.normalized = int(n_fft), int(n_fft) // 4, power, bool(normalized)
        self.register_buffer("window", torch.hann_window(self.n_fft), persistent=False)
        self.norm = (norm,) if isinstance(norm, str) else tuple(norm)

    def spec(self, x: torch.Tensor) -> torch.Tensor:
        """(..., T) -> (..., n_fft / 2 + 1, frames), torch.stft(center=False) layout."""
        w = self.window.to(x.dtype)
        fr = x.unfold(-1, self.n_fft, self.hop) * w                     # (..., frames, n_fft): no padding (center=False)
        s = torch.fft.rfft(fr, dim=-1).transpose(-1, -2)
        if self.normalized:
            s = s / w.pow(2.0).sum().sqrt()
        if self.power is None:
            return s
        return s.abs() if self.power == 1.0 else s.abs().pow(self.power)

    def forward(self, x, y):
        x = self.spec(x)
        y = self.spec(y)
        distance = 0
        for norm in self.norm:
            distance = distance + mean_difference(y, x, norm)
        return distance


class EncodecAudioDistance(nn.Module):
    """rave/core.py:415-431."""

    def __init__(self, scales, spectral_distance) -> None:
        super().__init__()
        self.waveform_distance = WaveformDistance(norm="L1")
        self.spectral_distances = nn.ModuleList([spectral_distance(scale) for scale in scales])

    def forward(self, x, y):
        waveform_distance = self.waveform_distance(x, y)
        spectral_distance = 0
        for dist in self.spectral_distances:
            spectral_distance = spectral_distance + dist(x, y)
        return {"waveform_distance": waveform_distance, "spectral_distance": spectral_distance}
