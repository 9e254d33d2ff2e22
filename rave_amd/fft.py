"""Batched 1-D real FFTs on rocFFT through the hipFFT C API of the library PyTorch already loaded
(torch/lib/libhipfft.so), called directly so that the frames / gradient buffers -- scratch tensors we own -- are
transformed without torch.fft's defensive input clones (one per R2C, two per C2R: 60 device copies per v2 step).
Plans are cached per (n_fft, batch, device); the stream is set per call.  If the library cannot be loaded the
functions use torch.fft (same rocFFT kernels, plus the clones)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Tuple

import torch

_HIPFFT_R2C, _HIPFFT_C2R = 0x2a, 0x2c
_lib = None
_plans: Dict[Tuple[int, int, int, int], int] = {}


def _load():
    global _lib
    if _lib is None:
        try:
            lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libhipfft.so"))
            lib.hipfftPlanMany.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                           C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]
            lib.hipfftSetStream.argtypes = [C.c_void_p, C.c_void_p]
            lib.hipfftExecR2C.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            lib.hipfftExecC2R.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            for f in (lib.hipfftPlanMany, lib.hipfftSetStream, lib.hipfftExecR2C, lib.hipfftExecC2R):
                f.restype = C.c_int
            _lib = lib
        except (OSError, AttributeError):
            _lib = False
    return _lib


def _plan(kind: int, n: int, batch: int, device: torch.device) -> int:
    key = (kind, n, batch, device.index or 0)
    p = _plans.get(key)
    if p is None:
        h = C.c_void_p()
        nn = (C.c_int * 1)(n)
        idist, odist = (n, n // 2 + 1) if kind == _HIPFFT_R2C else (n // 2 + 1, n)
        rc = _lib.hipfftPlanMany(C.byref(h), 1, nn, None, 1, idist, None, 1, odist, kind, batch)
        if rc != 0:
            raise RuntimeError(f"hipfftPlanMany failed ({rc}) for n={n} batch={batch}")
        p = _plans[key] = h.value
    return p


def rfft_last(frames: torch.Tensor) -> torch.Tensor:
    """rfft over the last dim of a contiguous fp32 GPU tensor; the input buffer may be overwritten."""
    n = frames.shape[-1]
    if not _load() or n % 2:
        return torch.fft.rfft(frames, dim=-1)
    batch = frames.numel() // n
    out = torch.empty(frames.shape[:-1] + (n // 2 + 1,), device=frames.device, dtype=torch.complex64)
    with torch.cuda.device(frames.device):
        p = _plan(_HIPFFT_R2C, n, batch, frames.device)
        _lib.hipfftSetStream(p, torch.cuda.current_stream().cuda_stream)
        rc = _lib.hipfftExecR2C(p, frames.data_ptr(), out.data_ptr())
    if rc != 0:
        raise RuntimeError(f"hipfftExecR2C failed ({rc})")
    return out


def irfft_last_unnormalized(spec: torch.Tensor, n: int) -> torch.Tensor:
    """Unnormalised C2R over the last dim (== torch.fft.irfft(spec, n, norm="forward")); spec is overwritten."""
    if not _load() or n % 2 or spec.shape[-1] != n // 2 + 1:
        return torch.fft.irfft(spec, n=n, dim=-1, norm="forward")
    batch = spec.numel() // spec.shape[-1]
    out = torch.empty(spec.shape[:-1] + (n,), device=spec.device, dtype=torch.float32)
    with torch.cuda.device(spec.device):
        p = _plan(_HIPFFT_C2R, n, batch, spec.device)
        _lib.hipfftSetStream(p, torch.cuda.current_stream().cuda_stream)
        rc = _lib.hipfftExecC2R(p, spec.data_ptr(), out.data_ptr())
    if rc != 0:
        raise RuntimeError(f"hipfftExecC2R failed ({rc})")
    return out
