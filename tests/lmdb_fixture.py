"""TEST INFRASTRUCTURE ONLY: writes an LMDB 0.9 data file (liblmdb's on-disk format as restated in rave_amd/lmdb_reader.py) holding
``'%08d'`` -> serialised AudioExample records the way scripts/preprocess.py:139-158 does, without liblmdb (not installed here).
Values larger than a quarter page go to overflow pages (F_BIGDATA), as liblmdb places them; ``fanout`` bounds the nodes per page
so that small fixtures still grow a multi-level tree."""
import os
import struct

PS = 4096
MAGIC = 0xBEEFC0DE


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _ld(fno, payload):
    return _varint((fno << 3) | 2) + _varint(len(payload)) + payload


def audio_example(pcm_bytes, channels, sr=44100, metadata=None, order=(1, 2, 3, 4)):
    """AudioExample{buffers: {'waveform': AudioBuffer{shape, sampling_rate, data, precision = INT16}}, metadata}; ``order`` =
    the field numbers given to (shape, sampling_rate, data, precision): the reader must not depend on them."""
    f_shape, f_sr, f_data, f_prec = order
    shape = b"".join(_varint(v) for v in (channels, len(pcm_bytes) // 2 // channels))
    buf = _ld(f_shape, shape) + _varint(f_sr << 3) + _varint(sr) + _ld(f_data, pcm_bytes) + _varint(f_prec << 3) + _varint(0)
    msg = _ld(1, _ld(1, b"waveform") + _ld(2, buf))
    for k, v in (metadata or {}).items():
        msg += _ld(2, _ld(1, k.encode()) + _ld(2, v.encode()))
    return msg


def write_lmdb(dirpath, records, fanout=5, txnid=7):
    """``records``: list of (key bytes, value bytes), any order.  Pages: 0 / 1 meta, then overflow runs, leaves, branches."""
    os.makedirs(dirpath, exist_ok=True)
    records = sorted(records)
    pages = {}                                  # pgno -> bytes
    next_pg = [2]
    n_overflow = 0

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p

    def page(pgno, flags, nodes):
        """nodes: list of node byte strings; packed from the END of the page, pointers from the start (as liblmdb)."""
        body = bytearray(PS)
        upper = PS
        ptrs = []
        for nd in nodes:
            nd = nd + b"\0" * (len(nd) & 1)       # nodes are 2-byte aligned
            upper -= len(nd)
            body[upper:upper + len(nd)] = nd
            ptrs.append(upper)
        lower = 16 + 2 * len(ptrs)
        assert lower <= upper, "page overflow in the fixture writer"
        struct.pack_into("<QHHHH", body, 0, pgno, 0, flags, lower, upper)
        struct.pack_into(f"<{len(ptrs)}H", body, 16, *ptrs)
        pages[pgno] = bytes(body)

    # leaves
    leaf_nodes = []
    for k, v in records:
        if len(v) > PS // 4:
            npg = (16 + len(v) + PS - 1) // PS
            first = alloc(npg)
            n_overflow += npg
            blob = bytearray(npg * PS)
            struct.pack_into("<QHHI", blob, 0, first, 0, 0x04, npg)
            blob[16:16 + len(v)] = v
            for i in range(npg):
                pages[first + i] = bytes(blob[i * PS:(i + 1) * PS])
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0x01, len(k)) + k + struct.pack("<Q", first)
        else:
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        leaf_nodes.append((k, node))
    level = []                                  # (first key, pgno)
    n_leaf = n_branch = 0
    for i in range(0, len(leaf_nodes), fanout):
        grp = leaf_nodes[i:i + fanout]
        pg = alloc()
        page(pg, 0x02, [nd for _, nd in grp])
        level.append((grp[0][0], pg))
        n_leaf += 1
    depth = 1 if level else 0
    while len(level) > 1:
        up = []
        for i in range(0, len(level), fanout):
            grp = level[i:i + fanout]
            pg = alloc()
            nodes = []
            for j, (k, child) in enumerate(grp):
                key = b"" if j == 0 else k
                nodes.append(struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, len(key)) + key)
            page(pg, 0x01, nodes)
            up.append((grp[0][0], pg))
            n_branch += 1
        level = up
        depth += 1
    root = level[0][1] if level else (1 << 64) - 1
    last = next_pg[0] - 1

    def meta(pgno, txn, root_, entries):
        body = bytearray(PS)
        struct.pack_into("<QHHHH", body, 0, pgno, 0, 0x08, 0, 0)
        struct.pack_into("<IIQQ", body, 16, MAGIC, 1, 0, 1 << 30)
        struct.pack_into("<IHHQQQQQ", body, 40, PS, 0, 0, 0, 0, 0, 0, (1 << 64) - 1)                      # free DB (empty)
        struct.pack_into("<IHHQQQQQ", body, 88, 0, 0, depth if entries else 0, n_branch, n_leaf, n_overflow, entries, root_)
        struct.pack_into("<QQ", body, 136, last, txn)
        return bytes(body)

    pages[0] = meta(0, txnid - 1, (1 << 64) - 1, 0)        # the OLDER meta: an empty database -- must not be the one read
    pages[1] = meta(1, txnid, root, len(records))
    with open(os.path.join(dirpath, "data.mdb"), "wb") as f:
        for p in range(next_pg[0]):
            f.write(pages.get(p, b"\0" * PS))
    return dict(depth=depth, leaves=n_leaf, branches=n_branch, overflow=n_overflow)
