"""Per-kernel register / LDS / scratch usage read from the gfx950 code objects embedded in librave_hip.so.

Parses the clang offload bundles in the .so, the AMDGPU metadata note (msgpack) of each device ELF, and prints one
line per kernel.  `scratch_kernels()` is what tests/test_abi_and_host.py uses to pin "no scratch in the MFMA kernels"
(a compiler-demoted accumulator array once cost 4x and went unnoticed).

    python tools/kernel_resources.py [pattern]
"""
import struct
import sys
from pathlib import Path

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_elfs(blob: bytes):
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + 1)


def _metadata(elf: bytes):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        h = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, h + 4)
        if sh_type != 7:                                   # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, h + 0x18)
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and ntype == 32:
                return msgpack.unpackb(desc, raw=False, strict_map_key=False)
    return None


def kernels(so_path=None):
    so_path = Path(so_path or Path(__file__).resolve().parents[1] / "rave_amd" / "librave_hip.so")
    out = []
    for elf in _device_elfs(so_path.read_bytes()):
        md = _metadata(elf)
        for k in (md or {}).get("amdhsa.kernels", []):
            out.append(dict(name=k[".name"], vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0),
                            sgpr=k.get(".sgpr_count", 0), lds=k.get(".group_segment_fixed_size", 0),
                            scratch=k.get(".private_segment_fixed_size", 0),
                            vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0)))
    return out


def scratch_kernels(patterns=("conv_x6_kernel", "wgrad_x6_kernel", "conv_igemm_dma_kernel", "wgrad_dma_kernel", "conv2d_x6_kernel",
                             "wgrad2d_x6_kernel", "unit_x6_kernel", "conv2d_smallm_kernel", "conv2d_smallc_fwd_kernel")):
    return [k for k in kernels() if any(p in k["name"] for p in patterns) and (k["scratch"] or k["vgpr_spill"])]


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    ks = [k for k in kernels() if pat in k["name"]]
    for k in sorted(ks, key=lambda k: k["name"]):
        print(f"{k['vgpr']:4d} vgpr {k['agpr']:4d} agpr {k['sgpr']:4d} sgpr {k['lds']:7d} lds "
              f"{k['scratch']:6d} scratch {k['vgpr_spill']:3d} spill  {k['name'][:110]}")
    print(f"{len(ks)} kernels, {sum(1 for k in ks if k['scratch'])} with scratch")
