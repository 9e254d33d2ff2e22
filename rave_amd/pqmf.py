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
import os

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


_PROTOTYPES = {}


def get_prototype(atten, M, N=None):
    """rave/pqmf.py:83-89.  The design (about 100 firwin calls inside scipy's fmin, several seconds) is a pure function
    of its arguments: the result is memoised per process."""
    key = (float(atten), int(M), None if N is None else int(N))
    if key not in _PROTOTYPES:
        from scipy.optimize import fmin
        wc = fmin(lambda w: loss_wc(w, atten, M, N), 1 / M, disp=0)[0]
        _PROTOTYPES[key] = kaiser_filter(wc, atten, N)
    return _PROTOTYPES[key].copy()


def get_qmf_bank(h, n_band):
    """rave/pqmf.py:32-52."""
    k = torch.arange(n_band).reshape(-1, 1)
    N = h.shape[-1]
    t = torch.arange(-(N // 2), N // 2 + 1)
    p = (-1) ** k * math.pi / 4
    mod = torch.cos((2 * k + 1) * math.pi / (2 * n_band) * t + p)
    return 2 * h * mod


def fold_tables(h: torch.Tensor, hk: torch.Tensor, tol: float = 5e-6):
    """Operands of the folded fast form (rave_amd/csrc/pqmf_fold.hip) for a 16-band bank:
    the signed prototype hs[tau] = (-1)^(tau/32) h[tau] (zero padded to 384) and Cm[k][m] = 2 cos((2k+1) pi/32 (m - N/2)
    + (-1)^k pi/4), computed in float64 from the module's own ``h`` buffer.  Returns ``(tab, lpad)`` -- or None when the
    stored bank ``hk`` (what forward_conv / inverse_conv hold) is NOT the closed-form cosine-modulated bank of ``h`` to
    ``tol`` relative L2 (a checkpoint with edited filters, another band count, a longer prototype): the direct-form
    kernels, exact w.r.t. the stored taps, are used then."""
    n = h.numel()
    if hk.dim() != 2 or hk.shape[0] != 16 or hk.shape[1] != 512 or n > 384 or n % 2 == 0:
        return None
    h64 = h.detach().double().cpu()
    tau = torch.arange(384)
    hs = torch.zeros(384, dtype=torch.float64)
    hs[:n] = h64
    hs = hs * ((-1.0) ** (tau // 32))
    k = torch.arange(16, dtype=torch.float64).reshape(-1, 1)
    m = torch.arange(32, dtype=torch.float64).reshape(1, -1)
    cm = 2 * torch.cos((2 * k + 1) * math.pi / 32 * (m - n // 2) + ((-1.0) ** k) * math.pi / 4)
    lpad = (512 - n) // 2
    closed = torch.zeros(16, 512, dtype=torch.float64)
    closed[:, lpad:lpad + n] = hs[None, :n] * cm[:, tau[:n] % 32]
    ref = hk.detach().double().cpu()
    err = float((closed - ref).norm() / ref.norm())
    if not err < tol:
        return None
    return torch.cat([hs, cm.reshape(-1)]).float(), lpad


def _fold_enabled() -> bool:
    return os.environ.get("RH_PQMF_FOLD", "1") != "0"


class _FoldMixin:
    """Lazily builds (and caches per device) the folded-form operands; ``None`` = use the direct-form kernels."""

    def _fold(self, w_fwd: torch.Tensor):
        if not _fold_enabled() or self.n_band != 16:
            return None
        key = (w_fwd.data_ptr(), w_fwd._version, str(w_fwd.device))
        cache = getattr(self, "_fold_cache", None)
        if cache is not None and w_fwd.is_cuda and torch.cuda.is_current_stream_capturing():
            return cache[1]        # a hipGraph is being recorded: no host round trip (refresh_host_caches ran before)
        if cache is None or cache[0] != key:
            # the stored conv weight is hk plus the make_odd zero tap: compare the bank it actually holds
            bank = w_fwd.detach().reshape(w_fwd.shape[0], -1)[:, :512]
            ft = fold_tables(self.h, bank)
            cache = (key, None if ft is None else (ft[0].to(w_fwd.device), ft[1]))
            self._fold_cache = cache
        return cache[1]


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


class CachedPQMF(_FoldMixin, nn.Module):
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
        return ops.pqmf_analysis(x, self.forward_conv.weight, self.forward_conv._pad, self._fold(self.forward_conv.weight))

    def inverse(self, x):
        """(B*C, 16, N) -> (B*C, 1, 16 N); sign flip, x16, band reversal, interleave fused
        (rave/pqmf.py:285-294)."""
        if self.n_band == 1:
            return x
        # (the synthesis bank is the time-reversed analysis bank: one check of forward_conv covers both; an edited
        # inverse_conv alone is caught by comparing it with the flipped forward bank)
        fold = self._fold(self.forward_conv.weight)
        if fold is not None and not self._inverse_matches():
            fold = None
        return ops.pqmf_synthesis(x, self.inverse_conv.weight, self.inverse_conv._pad, fold)

    def refresh_host_caches(self) -> None:
        """Re-validate the host-side decisions that depend on parameter VALUES (is the stored bank the closed-form
        one?) -- called before a hipGraph capture, during which no device-to-host copy is possible."""
        if self.n_band == 16 and self._fold(self.forward_conv.weight) is not None:
            self._inverse_matches()

    def _inverse_matches(self) -> bool:
        w = self.inverse_conv.weight
        key = (w.data_ptr(), w._version)
        c = getattr(self, "_inv_ok", None)
        if c is not None and w.is_cuda and torch.cuda.is_current_stream_capturing():
            return c[1]
        if c is None or c[0] != key:
            hk = self.forward_conv.weight.detach().reshape(16, -1)[:, :512]
            m = hk.shape[0]
            want = make_odd(hk.flip(-1).reshape(m, -1, m).permute(2, 0, 1))
            ok = bool(torch.equal(want.to(w.device), w.detach()))
            c = self._inv_ok = (key, ok)
        return c[1]


class PQMF(_FoldMixin, nn.Module):
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
        return ops.pqmf_analysis(x, self._w_fwd, (half, half), self._fold(self._w_fwd))

    def inverse(self, x):
        if x.ndim == 2:
            if self.n_channels == 1:
                return self.inverse(x[0]).unsqueeze(0)
            x = x.split(self.n_channels, -2)
            return torch.stack([self.inverse(xi) for xi in x])
        if self.n_band == 1:
            return x
        taps = self._w_inv.shape[-1] - 1          # 32 polyphase taps (+1 zero from make_odd)
        return ops.pqmf_synthesis(x, self._w_inv, (taps // 2 - 1, taps // 2 + 1), self._fold(self._w_fwd))
