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
