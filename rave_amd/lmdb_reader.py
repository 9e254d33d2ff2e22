"""Read-only reader of the on-disk dataset container of the reference: the LMDB environment written by
``scripts/preprocess.py:139-158`` -- key ``'%08d'`` -> serialised ``udls`` ``AudioExample`` protobuf whose ``buffers['waveform']``
holds one chunk of int16 PCM -- as consumed by ``rave.dataset.AudioDataset`` (rave/dataset.py:33-84: every key in cursor order,
``np.frombuffer(buffer.data, int16)`` reshaped to (n_channels, -1)).

Both formats live in third-party packages that are NOT vendored under /root/reference and are not installed here (no network):

* ``lmdb`` (py-lmdb over liblmdb 0.9.x, pulled in by ``udls>=1.0.1``, requirements.txt:13).  The file format restated here is
  liblmdb 0.9's (mdb.c): 4 KiB-or-larger pages with a 16-byte header (pgno u64, pad u16, flags u16, then lower / upper u16 -- or
  the overflow page count u32); two meta pages (magic 0xBEEFC0DE, version 1, two 48-byte ``MDB_db`` records: free list and main
  DB, last page, transaction id -- the meta with the larger transaction id is current; the page size is the free DB's ``md_pad``);
  B+tree of branch pages (nodes: 48-bit child page number in lo / hi / flags, key) and leaf pages (nodes: 32-bit data size in lo
  / hi, flags, key size, key, then the data inline or -- F_BIGDATA -- the u64 number of the first of a run of overflow pages whose
  payload starts 16 bytes in).  Little endian, as written on the x86-64 / aarch64 hosts the reference runs on.
* ``udls.generated.AudioExample`` (proto3): ``map<string, AudioBuffer> buffers = 1; map<string, string> metadata = 2`` with
  ``AudioBuffer {repeated int32 shape; int32 sampling_rate; bytes data; Precision precision}`` and ``Precision.INT16 = 0``
  (call sites: scripts/preprocess.py:139-158, rave/dataset.py:59,70-76).  The parser below decodes the protobuf wire format
  generically and does not depend on AudioBuffer's field NUMBERS: the PCM is the (only large) length-delimited field of the
  buffer, the small varints are sampling rate / precision / shape.

PARITY UNPINNED: neither library is available in this image, so this reader is checked against files produced by the writer in
tests/lmdb_fixture.py (same restated format), not against liblmdb itself.  Host I/O, not on the GPU hot path: it runs once, and
what it returns is moved into HBM for ``rave_amd.data.GpuBatchFeed``.
"""
from __future__ import annotations

import mmap
import os
import struct
from typing import Dict, Iterator, List, Optional, Tuple

MDB_MAGIC = 0xBEEFC0DE
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
PAGEHDRSZ = 16
P_INVALID = (1 << 64) - 1


class LmdbFormatError(RuntimeError):
    pass


class LmdbReader:
    """``LmdbReader(path)``: ``path`` is the environment directory (holding ``data.mdb``, py-lmdb's default ``subdir=True``) or
    the data file itself.  ``keys()`` / ``items()`` walk the main database in key order (what ``txn.cursor()`` yields),
    ``get(key)`` descends the tree.  Values are returned as ``memoryview``s into the mapping (zero copy) -- copy before close()."""

    def __init__(self, path: str) -> None:
        if os.path.isdir(path):
            path = os.path.join(path, "data.mdb")
        self._f = open(path, "rb")
        size = os.fstat(self._f.fileno()).st_size
        if size < 2 * 512:
            raise LmdbFormatError(f"{path}: too small to be an LMDB data file")
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._mv = memoryview(self._m)
        self._read_meta(path)

    def close(self) -> None:
        """Unmaps the file -- unless value views handed out by items() / get() are still alive: they keep the mapping, which
        then goes away with the last of them."""
        try:
            self._mv.release()
            self._m.close()
        except BufferError:
            pass
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- meta pages
    def _meta_at(self, off: int):
        flags = struct.unpack_from("<H", self._m, off + 10)[0]
        magic, version = struct.unpack_from("<II", self._m, off + PAGEHDRSZ)
        if magic != MDB_MAGIC or not flags & P_META:
            return None
        base = off + PAGEHDRSZ + 8 + 8 + 8            # magic + version, mm_address, mm_mapsize
        dbs = []
        for i in range(2):                            # FREE_DBI, MAIN_DBI
            pad, dflags, depth, branch, leaf, ovf, entries, root = struct.unpack_from("<IHHQQQQQ", self._m, base + 48 * i)
            dbs.append(dict(pad=pad, flags=dflags, depth=depth, branch_pages=branch, leaf_pages=leaf, overflow_pages=ovf,
                            entries=entries, root=root))
        last_pg, txnid = struct.unpack_from("<QQ", self._m, base + 96)
        return dict(version=version, dbs=dbs, last_pg=last_pg, txnid=txnid)

    def _read_meta(self, path: str) -> None:
        m0 = self._meta_at(0)
        if m0 is None:
            raise LmdbFormatError(f"{path}: no LMDB meta page (magic 0x{MDB_MAGIC:X}) at offset 0")
        psize = m0["dbs"][0]["pad"]
        if psize < 512 or psize & (psize - 1) or 2 * psize > len(self._m):
            raise LmdbFormatError(f"{path}: implausible page size {psize}")
        m1 = self._meta_at(psize)
        meta = m0 if m1 is None or m0["txnid"] >= m1["txnid"] else m1
        if meta["version"] != 1:
            raise LmdbFormatError(f"{path}: LMDB data version {meta['version']} (only version 1 = liblmdb 0.9.x is read)")
        self.page_size = psize
        self.meta = meta
        main = meta["dbs"][1]
        if main["flags"] & 0x04:                      # MDB_DUPSORT
            raise LmdbFormatError(f"{path}: the main database has duplicate keys (MDB_DUPSORT): not a preprocess.py dataset")
        self.entries = int(main["entries"])
        self.root = int(main["root"])
        self.depth = int(main["depth"])

    # ---- pages and nodes
    def _page(self, pgno: int) -> Tuple[int, int, int, int]:
        off = pgno * self.page_size
        if pgno == P_INVALID or off + PAGEHDRSZ > len(self._m):
            raise LmdbFormatError(f"page {pgno} lies outside the file")
        _, _, flags, lower, upper = struct.unpack_from("<QHHHH", self._m, off)
        return off, flags, lower, upper

    def _nodes(self, off: int, lower: int) -> List[int]:
        n = (lower - PAGEHDRSZ) >> 1
        return list(struct.unpack_from(f"<{n}H", self._m, off + PAGEHDRSZ)) if n > 0 else []

    def _leaf_value(self, off: int, nptr: int):
        lo, hi, nflags, ksize = struct.unpack_from("<HHHH", self._m, off + nptr)
        dsize = lo | (hi << 16)
        kstart = off + nptr + 8
        key = bytes(self._m[kstart:kstart + ksize])
        if nflags & (F_SUBDATA | F_DUPDATA):
            raise LmdbFormatError("sub-database / duplicate-data node in the main database: not a preprocess.py dataset")
        dstart = kstart + ksize
        if nflags & F_BIGDATA:
            ovf = struct.unpack_from("<Q", self._m, dstart)[0]
            ooff, oflags, _, _ = self._page(ovf)
            if not oflags & P_OVERFLOW:
                raise LmdbFormatError(f"page {ovf} referenced as overflow data is not an overflow page")
            dstart = ooff + PAGEHDRSZ
        if dstart + dsize > len(self._m):
            raise LmdbFormatError("value runs past the end of the file")
        return key, self._mv[dstart:dstart + dsize]

    def _branch_child(self, off: int, nptr: int) -> Tuple[bytes, int]:
        lo, hi, nflags, ksize = struct.unpack_from("<HHHH", self._m, off + nptr)
        kstart = off + nptr + 8
        return bytes(self._m[kstart:kstart + ksize]), lo | (hi << 16) | (nflags << 32)

    def items(self) -> Iterator[Tuple[bytes, memoryview]]:
        """(key, value) pairs of the main database in key order."""
        if self.root == P_INVALID or self.entries == 0:
            return
        stack = [self.root]
        while stack:
            pg = stack.pop()
            off, flags, lower, _ = self._page(pg)
            ptrs = self._nodes(off, lower)
            if flags & P_LEAF:
                if flags & P_LEAF2:
                    raise LmdbFormatError("LEAF2 page in the main database: not a preprocess.py dataset")
                for nptr in ptrs:
                    yield self._leaf_value(off, nptr)
            elif flags & P_BRANCH:
                stack.extend(self._branch_child(off, nptr)[1] for nptr in reversed(ptrs))
            else:
                raise LmdbFormatError(f"page {pg}: neither branch nor leaf (flags 0x{flags:x})")

    def keys(self) -> List[bytes]:
        return [k for k, _ in self.items()]

    def get(self, key: bytes) -> Optional[memoryview]:
        if self.root == P_INVALID:
            return None
        pg = self.root
        for _ in range(64):
            off, flags, lower, _ = self._page(pg)
            ptrs = self._nodes(off, lower)
            if flags & P_LEAF:
                lo, hi = 0, len(ptrs)
                while lo < hi:                         # first node with key >= the wanted one
                    mid = (lo + hi) // 2
                    k, _ = self._leaf_key(off, ptrs[mid])
                    if k < key:
                        lo = mid + 1
                    else:
                        hi = mid
                if lo < len(ptrs):
                    k, v = self._leaf_value(off, ptrs[lo])
                    if k == key:
                        return v
                return None
            # branch: the last child whose separator key is <= the wanted key (node 0 has an empty key = minus infinity)
            child = self._branch_child(off, ptrs[0])[1]
            lo, hi = 1, len(ptrs)
            while lo < hi:
                mid = (lo + hi) // 2
                if self._branch_child(off, ptrs[mid])[0] <= key:
                    lo = mid + 1
                else:
                    hi = mid
            if lo - 1 >= 1:
                child = self._branch_child(off, ptrs[lo - 1])[1]
            pg = child
        raise LmdbFormatError("tree deeper than 64 levels")

    def _leaf_key(self, off: int, nptr: int):
        ksize = struct.unpack_from("<H", self._m, off + nptr + 6)[0]
        return bytes(self._m[off + nptr + 8:off + nptr + 8 + ksize]), None


# --------------------------------------------------------------------------- protobuf (wire format, schema-light)
def _varint(buf, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise LmdbFormatError("truncated protobuf varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise LmdbFormatError("malformed protobuf varint")


def _fields(buf) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message: value is an int (varint / fixed) or a memoryview (length-delimited)."""
    mv = memoryview(buf)
    pos, end = 0, len(mv)
    while pos < end:
        tag, pos = _varint(mv, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(mv, pos)
        elif wt == 1:
            v = int.from_bytes(mv[pos:pos + 8], "little"); pos += 8
        elif wt == 2:
            n, pos = _varint(mv, pos)
            if pos + n > end:
                raise LmdbFormatError("protobuf length-delimited field runs past its message")
            v = mv[pos:pos + n]; pos += n
        elif wt == 5:
            v = int.from_bytes(mv[pos:pos + 4], "little"); pos += 4
        else:
            raise LmdbFormatError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _map_entry(buf) -> Tuple[str, memoryview]:
    key, val = "", memoryview(b"")
    for fno, wt, v in _fields(buf):
        if fno == 1 and wt == 2:
            key = bytes(v).decode("utf-8", "replace")
        elif fno == 2 and wt == 2:
            val = v
    return key, val


def parse_audio_buffer(buf) -> Dict[str, object]:
    """``AudioExample.AudioBuffer`` -> dict(data=memoryview, small={field number: [ints]}): the PCM is the largest
    length-delimited field; every varint field (and every other length-delimited field of <= 64 bytes, read as packed
    varints) is kept under ``small`` -- sampling rate, precision and shape, whatever their field numbers."""
    blobs = []
    small: Dict[int, List[int]] = {}
    for fno, wt, v in _fields(buf):
        if wt == 2:
            blobs.append((fno, v))
        elif wt == 0:
            small.setdefault(fno, []).append(int(v))
    data = memoryview(b"")
    if blobs:
        big = max(range(len(blobs)), key=lambda i: len(blobs[i][1]))
        data = blobs[big][1]
        for i, (fno, v) in enumerate(blobs):
            if i != big and len(v) <= 64:
                small.setdefault(fno, []).extend(_packed(v))
    return dict(data=data, small=small)


def _packed(v) -> List[int]:
    out, pos = [], 0
    try:
        while pos < len(v):
            x, pos = _varint(v, pos)
            out.append(x)
    except IndexError:
        return []
    return out


def parse_audio_example(buf) -> Tuple[Dict[str, Dict[str, object]], Dict[str, str]]:
    """Serialised ``AudioExample`` -> (buffers: name -> parse_audio_buffer(...), metadata: str -> str)."""
    buffers: Dict[str, Dict[str, object]] = {}
    metadata: Dict[str, str] = {}
    for fno, wt, v in _fields(buf):
        if wt != 2:
            continue
        k, val = _map_entry(v)
        if fno == 1:
            buffers[k] = parse_audio_buffer(val)
        elif fno == 2:
            metadata[k] = bytes(val).decode("utf-8", "replace")
    return buffers, metadata
