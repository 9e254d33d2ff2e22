"""CPU tests of the dataset-container reader (rave_amd/lmdb_reader.py, rave_amd/data.load_lmdb_dataset)
against stores written by tests/lmdb_fixture.py in the layout of scripts/preprocess.py:139-158.  liblmdb / udls are not installed
in this image: the format is restated, parity with the C library is unpinned (said in the reader's header)."""
import os

import numpy as np
import pytest
import torch

from lmdb_fixture import audio_example, write_lmdb
from rave_amd.lmdb_reader import LmdbFormatError, LmdbReader, parse_audio_example


def _chunks(n_items, channels, length, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(-32768, 32767, (channels, length), dtype=np.int16) for _ in range(n_items)]


@pytest.mark.parametrize("n_items,fanout", [(1, 5), (4, 5), (37, 3), (130, 5)])
def test_reader_walks_branch_leaf_and_overflow_pages_in_key_order(tmp_path, n_items, fanout):
    chunks = _chunks(n_items, 1, 5000)
    recs = [(f"{i:08d}".encode(), audio_example(c.tobytes(), 1)) for i, c in enumerate(chunks)]
    recs.append((b"zz_small", b"inline value"))                    # a value stored inline in its leaf node
    info = write_lmdb(str(tmp_path), recs[::-1], fanout=fanout)     # (the writer sorts: insertion order does not matter)
    if n_items >= 37:
        assert info["depth"] >= 3 and info["branches"] >= 2 and info["overflow"] > n_items
    with LmdbReader(str(tmp_path)) as db:
        assert db.entries == n_items + 1 and db.page_size == 4096
        keys = db.keys()
        assert keys == sorted(k for k, _ in recs)
        for (k, v), (kk, vv) in zip(sorted(recs), db.items()):
            assert k == kk and bytes(vv) == v
        assert bytes(db.get(b"zz_small")) == b"inline value"
        assert bytes(db.get(f"{n_items - 1:08d}".encode())) == dict(recs)[f"{n_items - 1:08d}".encode()]
        assert db.get(b"00000000x") is None and db.get(b"") is None and db.get(b"zzzz") is None


@pytest.mark.parametrize("order", [(1, 2, 3, 4), (4, 1, 2, 3), (2, 3, 1, 7)])
def test_audio_example_parser_does_not_depend_on_buffer_field_numbers(order):
    pcm = _chunks(1, 2, 777, seed=3)[0]
    msg = audio_example(pcm.tobytes(), 2, sr=48000, metadata={"path": "a/b.wav", "length": "1.5"}, order=order)
    buffers, meta = parse_audio_example(msg)
    assert meta == {"path": "a/b.wav", "length": "1.5"}
    got = np.frombuffer(buffers["waveform"]["data"], dtype=np.int16).reshape(2, -1)
    assert np.array_equal(got, pcm)
    small = buffers["waveform"]["small"]
    assert sorted(v for vs in small.values() for v in vs) == sorted([2, 777, 48000, 0])


def test_load_lmdb_dataset_is_what_audio_dataset_serves(tmp_path):
    """rave/dataset.py:66-84: item i = frombuffer(buffers['waveform'].data, int16).reshape(n_channels, -1) of the i-th key."""
    from rave_amd.data import load_lmdb_dataset
    chunks = _chunks(9, 2, 4096, seed=5)
    recs = [(f"{i:08d}".encode(), audio_example(c.tobytes(), 2)) for i, c in enumerate(chunks)]
    recs.append((b"00000009", audio_example(_chunks(1, 2, 100, seed=9)[0].tobytes(), 2)))       # an odd-length straggler
    write_lmdb(str(tmp_path), recs, fanout=4)
    with open(os.path.join(str(tmp_path), "metadata.yaml"), "w") as f:
        f.write("channels: 2\nlazy: false\nn_seconds: 1.0\nsr: 48000\n")
    pcm, info = load_lmdb_dataset(str(tmp_path), device="cpu")
    assert pcm.dtype == torch.int16 and tuple(pcm.shape) == (9, 2, 4096)
    assert info == dict(sr=48000, channels=2, lazy=False, n_items=9, length=4096, dropped=1)
    for i, c in enumerate(chunks):
        assert np.array_equal(pcm[i].numpy(), c)
    with open(os.path.join(str(tmp_path), "metadata.yaml"), "w") as f:
        f.write("channels: 2\nlazy: true\nsr: 48000\n")
    with pytest.raises(RuntimeError, match="lazy"):
        load_lmdb_dataset(str(tmp_path), device="cpu")


def test_reader_refuses_what_is_not_an_lmdb_file(tmp_path):
    p = os.path.join(str(tmp_path), "data.mdb")
    with open(p, "wb") as f:
        f.write(b"\0" * 8192)
    with pytest.raises(LmdbFormatError):
        LmdbReader(str(tmp_path))
    write_lmdb(str(tmp_path), [])
    with LmdbReader(str(tmp_path)) as db:
        assert db.entries == 0 and db.keys() == [] and db.get(b"00000000") is None
