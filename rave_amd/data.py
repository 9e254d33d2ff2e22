"""GPU-side training data feed (SURVEY.md section 8f "next" #4): the default transform chain of
``rave.dataset.get_dataset`` (rave/dataset.py:218-229,246) -- int16 -> float32, RandomCrop, RandomApply(random
all-pass "phase mangle", p = .8), Dequantize(16) -- for a whole minibatch in one HIP launch (rh_feed_batch_i16_f32)
on PCM that is already resident in HBM (288 GB hold ~900 h of 44.1 kHz mono int16).

The random draws follow the reference's distributions (uniform item / crop point, log-uniform pole frequency in
[20, 2000] Hz at radius .99, U[0,1) dequantisation noise) but not its RNG streams; ``draws`` can be injected, which is
how the parity test compares with scipy.signal.lfilter.  ``load_lmdb_dataset`` reads the reference's on-disk container (the
lmdb / ``AudioExample`` store of scripts/preprocess.py, rave_amd/lmdb_reader.py) into the int16 tensor this class samples from.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import _lib as L


def pole_to_z_filter(omega: float, amplitude: float = .9):
    """rave/dataset.py:290-294."""
    z0 = amplitude * np.exp(1j * omega)
    a = [1, -2 * np.real(z0), abs(z0) ** 2]
    b = [abs(z0) ** 2, -2 * np.real(z0), 1]
    return b, a


def random_angle(rng: np.random.Generator, min_f=20, max_f=8000, sr=24000) -> float:
    """rave/dataset.py:283-288."""
    lo, hi = math.log(min_f), math.log(max_f)
    return 2 * math.pi * math.exp(rng.random() * (hi - lo) + lo) / sr


class GpuBatchFeed:
    """``pcm``: int16 tensor (n_items, n_channels, length) on the GPU.  ``sample`` returns a (B, n_channels, n_signal)
    float32 minibatch ready for ``RAVE.training_step``."""

    def __init__(self, pcm: torch.Tensor, sr: int = 44100, seed: int = 0, p_mangle: float = .8, bit_depth: int = 16):
        if pcm.dtype != torch.int16 or pcm.dim() != 3 or not pcm.is_cuda:
            raise RuntimeError("GpuBatchFeed: pcm must be an int16 (items, channels, length) tensor on the GPU")
        self.pcm = pcm.contiguous()
        self.sr, self.p_mangle, self.bit_depth = sr, p_mangle, bit_depth
        self.rng = np.random.default_rng(seed)
        self.gen = torch.Generator(device=pcm.device).manual_seed(seed)

    def draw(self, batch: int, n_signal: int):
        n_items, _, length = self.pcm.shape
        if n_signal > length:
            raise RuntimeError("GpuBatchFeed: n_signal longer than the stored items")
        items = self.rng.integers(0, n_items, batch)
        in_points = self.rng.integers(0, length - n_signal + 1, batch)                  # randint(0, L - n) inclusive
        angles = [random_angle(self.rng, 20, 2000, self.sr) if self.rng.random() < self.p_mangle else None
                  for _ in range(batch)]
        return items, in_points, angles

    def sample(self, batch: int, n_signal: int, draws=None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        items, in_points, angles = draws if draws is not None else self.draw(batch, n_signal)
        n_items, n_ch, length = self.pcm.shape
        rows = batch * n_ch
        off = np.empty(rows, dtype=np.int64)
        coef = np.full((rows, 5), np.nan, dtype=np.float64)
        for b in range(batch):
            for c in range(n_ch):
                r = b * n_ch + c
                off[r] = (int(items[b]) * n_ch + c) * length + int(in_points[b])
                if angles[b] is not None:
                    bb, aa = pole_to_z_filter(angles[b], .99)
                    coef[r] = [bb[0], bb[1], bb[2], aa[1], aa[2]]
        dev = self.pcm.device
        off_d = torch.from_numpy(off).to(dev)
        coef_d = torch.from_numpy(coef).to(dev)
        if noise is None:
            noise = torch.rand(rows, n_signal, device=dev, generator=self.gen)
        noise = noise.to(dev, torch.float32).contiguous()
        out = torch.empty(batch, n_ch, n_signal, device=dev, dtype=torch.float32)
        L.check(L.lib.rh_feed_batch_i16_f32(L.ptr(self.pcm), L.ptr(off_d), L.ptr(coef_d), L.ptr(noise), rows, n_signal,
                                            self.bit_depth, L.ptr(out), L.stream()), "feed_batch")
        return out


def load_lmdb_dataset(db_path: str, n_channels: Optional[int] = None, audio_key: str = "waveform", device=None,
                      max_items: Optional[int] = None):
    """The dataset ``scripts/preprocess.py`` wrote (``db_path/data.mdb`` + ``db_path/metadata.yaml``) as ONE int16 tensor
    (items, channels, length) -- what ``rave.dataset.AudioDataset`` (rave/dataset.py:33-84) serves item by item:
    every key of the environment in cursor order, ``np.frombuffer(ae.buffers['waveform'].data, int16).reshape(channels, -1)``.
    Returns (pcm, info): ``pcm`` on ``device`` (None: host memory, pinned when a GPU is present) for ``GpuBatchFeed``;
    ``info`` = dict(sr, channels, lazy, n_items, length, dropped).  Chunks are of equal length by construction
    (scripts/preprocess.py:83-98 drops an incomplete tail); should a store hold other lengths, the items that differ from the
    most common length are dropped and counted.  ``lazy`` stores (file paths decoded by ffmpeg on the fly, rave/dataset.py:
    87-170) hold no PCM and are refused."""
    import os
    import yaml
    from .lmdb_reader import LmdbReader, parse_audio_example
    info = {}
    meta_path = os.path.join(db_path, "metadata.yaml")
    if os.path.exists(meta_path):
        with open(meta_path) as f:
            info = yaml.safe_load(f) or {}
    if info.get("lazy"):
        raise RuntimeError("rave_amd.data.load_lmdb_dataset: lazy dataset (audio decoded on the fly by ffmpeg): no PCM in the store")
    ch = int(n_channels or info.get("channels", 1))
    # two passes, one copy (a list of per-item copies + np.stack + pin_memory held the dataset three times): the first
    # pass only counts the items and their lengths, the second writes each item straight into its row of ONE preallocated
    # (pinned, when a GPU is present and the data stays on the host) int16 tensor
    def items(db):
        n = 0
        for k, v in db.items():
            buffers, _ = parse_audio_example(v)
            if audio_key not in buffers:
                continue
            raw = buffers[audio_key]["data"]
            if len(raw) % (2 * ch):
                raise RuntimeError(f"load_lmdb_dataset: item {k!r} holds {len(raw)} bytes, not a whole number of {ch}-channel int16 frames")
            yield raw
            n += 1
            if max_items is not None and n >= max_items:
                return

    with LmdbReader(db_path) as db:
        lengths = [len(raw) // (2 * ch) for raw in items(db)]
        if not lengths:
            raise RuntimeError(f"load_lmdb_dataset: no '{audio_key}' buffers found under {db_path}")
        common = max(set(lengths), key=lengths.count)
        n_keep = lengths.count(common)
        pcm = torch.empty(n_keep, ch, common, dtype=torch.int16, pin_memory=(device is None and torch.cuda.is_available()))
        view = pcm.numpy()
        row = 0
        for raw in items(db):
            if len(raw) // (2 * ch) == common:
                view[row] = np.frombuffer(raw, dtype=np.int16).reshape(ch, -1)
                row += 1
    if device is not None:
        pcm = pcm.to(device)
    keep, chunks = range(n_keep), lengths
    out = dict(sr=int(info.get("sr", 44100)), channels=ch, lazy=False, n_items=len(keep), length=common,
               dropped=len(chunks) - len(keep))
    return pcm, out
