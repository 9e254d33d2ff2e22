"""Drop-in for ``rave.pqmf.CachedPQMF`` (rave/pqmf.py:179-294) on the HIP PQMF kernels.

Same constructor ``(attenuation, n_band, polyphase=True, n_channels=1)`` and the same
``state_dict``: buffers ``hk (16,512)``, ``h (377,)``, parameters ``forward_conv.weight (16,1,513)``
and ``inverse_conv.weight (16,16,33)``.  The prototype filter design is init-time host work (scipy)
and follows rave/pqmf.py:55-89 line by line with the current scipy spelling (``firwin(fs=2*pi)``
== the pinned scipy 1.10 ``firwin(nyq=pi)``).

Deliberate deviation (documented in DESIGN.md): the reference registers the two conv weights as
trainable Parameters that belong to no optimizer (Appendix B #5); here they keep their names but
``requires_grad=False``, so the useless weight-gradient convolution is skipped.  Parameter
trajectories of encoder/decoder/discriminator are unaffected.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import cc, ops


def kaiser_filter(wc, atten, N=None):
    """rave/pqmf.py:55-70."""
    from scipy.signal import firwin, kaiserord
    N_, beta = kaiserord(atten, wc / np.pi)
    N_ = 2 * (N_ // 2) + 1
    N = N if N is not None else N_
    return firwin(N, wc, window=("kaiser", beta), scale=False, fs=2 * np.pi)


def loss_wc(wc, atten, M, N):
    """rave/pqmf.py:73-80."""
    h = kaiser_filter(wc, atten, N)
    g = np.convolve(h, h[::-1], "full")
    g = abs(g[g.shape[-1] // 2::2 * M][1:])
    return np.max(g)


def get_prototype(atten, M, N=None):
    """rave/pqmf.py:83-89."""
    from scipy.optimize import fmin
    wc = fmin(lambda w: loss_wc(w, atten, M, N), 1 / M, disp=0)[0]
    return kaiser_filter(wc, atten, N)


def get_qmf_bank(h, n_band):
    """rave/pqmf.py:32-52."""
    k = torch.arange(n_band).reshape(-1, 1)
    N = h.shape[-1]
    t = torch.arange(-(N // 2), N // 2 + 1)
    p = (-1) ** k * math.pi / 4
    mod = torch.cos((2 * k + 1) * math.pi / (2 * n_band) * t + p)
    return 2 * h * mod


def center_pad_next_pow_2(x):
    """rave/pqmf.py:20-23."""
    next_2 = 2 ** math.ceil(math.log2(x.shape[-1]))
    pad = next_2 - x.shape[-1]
    return nn.functional.pad(x, (pad // 2, pad // 2 + int(pad % 2)))


def make_odd(x):
    """rave/pqmf.py:26-29."""
    if not x.shape[-1] % 2:
        x = nn.functional.pad(x, (0, 1))
    return x


class CachedPQMF(nn.Module):
    def __init__(self, attenuation, n_band, polyphase=True, n_channels=1):
        super().__init__()
        if n_band != 16 and n_band != 1:
            raise NotImplementedError("rave_amd.pqmf: the HIP kernels implement the 16-band bank of the shipped configs")
        h = torch.from_numpy(get_prototype(attenuation, n_band)).float()
        hk = center_pad_next_pow_2(get_qmf_bank(h, n_band))
        self.register_buffer("hk", hk)
        self.register_buffer("h", h)
        self.n_band = n_band
        self.polyphase = polyphase
        self.n_channels = n_channels

        hkf = make_odd(self.hk).unsqueeze(1)
        hki = self.hk.flip(-1)
        m = self.hk.shape[0]
        hki = hki.reshape(m, -1, m).permute(2, 0, 1)           # "c (t m) -> m c t"
        hki = make_odd(hki)
        # same sub-module / parameter names as the reference (cc.Conv1d holders)
        self.forward_conv = cc.Conv1d(hkf.shape[1], hkf.shape[0], hkf.shape[2],
                                      padding=cc.get_padding(hkf.shape[-1]), stride=hkf.shape[0], bias=False)
        self.forward_conv.weight.data.copy_(hkf)
        self.inverse_conv = cc.Conv1d(hki.shape[1], hki.shape[0], hki.shape[-1],
                                      padding=cc.get_padding(hki.shape[-1]), bias=False)
        self.inverse_conv.weight.data.copy_(hki)
        self.forward_conv.weight.requires_grad_(False)
        self.inverse_conv.weight.requires_grad_(False)

    def script_cache(self):
        pass

    def forward(self, x):
        """(B*C, 1, T) -> (B*C, 16, T/16), reverse_half fused (rave/pqmf.py:279-283)."""
        if self.n_band == 1:
            return x
        return ops.pqmf_analysis(x, self.forward_conv.weight, self.forward_conv._pad)

    def inverse(self, x):
        """(B*C, 16, N) -> (B*C, 1, 16 N); sign flip, x16, band reversal, interleave fused
        (rave/pqmf.py:285-294)."""
        if self.n_band == 1:
            return x
        return ops.pqmf_synthesis(x, self.inverse_conv.weight, self.inverse_conv._pad)


class PQMF(nn.Module):
    """rave/pqmf.py:179-242, the non-cached filterbank (``polyphase_forward/inverse`` :92-134 or
    ``classic_forward/inverse`` :137-176; not bound by any shipped .gin, kept as the cross-check the reference
    itself uses).  state_dict: buffers ``hk``, ``h`` only.  Both analysis variants equal CachedPQMF.forward
    (same 512 taps, pad 256); both synthesis variants equal CachedPQMF.inverse advanced by one frame of 16 samples
    (the polyphase form pads 17 + 17 and crops two frames, :129-133), which on the HIP kernel is the pad pair
    (15, 17) instead of (16, 16).  The kernel operands are non-persistent buffers derived from ``hk``."""

    def __init__(self, attenuation, n_band, polyphase=True, n_channels=1):
        super().__init__()
        if n_band != 16 and n_band != 1:
            raise NotImplementedError("rave_amd.pqmf: the HIP kernels implement the 16-band bank of the shipped configs")
        if polyphase:
            power = math.log2(n_band)
            assert power == math.floor(power), "when using the polyphase algorithm, n_band must be a power of 2"
        h = torch.from_numpy(get_prototype(attenuation, n_band)).float()
        hk = center_pad_next_pow_2(get_qmf_bank(h, n_band))
        self.register_buffer("hk", hk)
        self.register_buffer("h", h)
        self.n_band = n_band
        self.polyphase = polyphase
        self.n_channels = n_channels
        m = hk.shape[0]
        self.register_buffer("_w_fwd", make_odd(hk).unsqueeze(1).contiguous(), persistent=False)
        self.register_buffer("_w_inv", make_odd(hk.flip(-1).reshape(m, -1, m).permute(2, 0, 1)).contiguous(),
                             persistent=False)

    def forward(self, x):
        if x.ndim == 2:
            return torch.stack([self.forward(x[i]) for i in range(x.shape[0])])
        if self.n_band == 1:
            return x
        half = self.hk.shape[-1] // 2
        return ops.pqmf_analysis(x, self._w_fwd, (half, half))

    def inverse(self, x):
        if x.ndim == 2:
            if self.n_channels == 1:
                return self.inverse(x[0]).unsqueeze(0)
            x = x.split(self.n_channels, -2)
            return torch.stack([self.inverse(xi) for xi in x])
        if self.n_band == 1:
            return x
        taps = self._w_inv.shape[-1] - 1          # 32 polyphase taps (+1 zero from make_odd)
        return ops.pqmf_synthesis(x, self._w_inv, (taps // 2 - 1, taps // 2 + 1))
