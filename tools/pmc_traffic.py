"""Assemble profiles/roundN_pmc_traffic.json from rocprofv3 PMC passes (.db files):

    python tools/pmc_traffic.py FETCH.db WRITE.db OUT.json [CALIB_FETCH.db CALIB_WRITE.db]

HBM bytes per launch of the kernel families bench.py prices = read_factor * FETCH_SIZE KB * 1024 + write_factor *
WRITE_SIZE KB * 1024.  The factors are CALIBRATED per access pattern when the two passes over tools/probe/fetch_calib (256
MiB per kernel: 16-byte loads, lane-consecutive dword loads, the 8-row dword gather of conv_x6_kernel's activation
loader, dword / 16-byte stores) are given; without them the guide's value for wide streaming reads (FETCH x 2, WRITE x 1)
is used and the result says so."""
import json, re, sqlite3, sys
from collections import defaultdict

# family -> (kernel-name prefixes, read pattern, write pattern)
FAMILIES = {
    "pqmf": (("pqmf_",), "read_dword", "write_x4"),
    "conv_x6(fwd+dgrad)": (("conv_x6_kernel", "unit_x6_kernel"), "read_rows8", "write_dword"),
    "conv_f32(fwd+dgrad)": (("conv_igemm_dma_kernel", "conv_igemm_kernel"), "read_x4", "write_dword"),
    "wgrad_x6": (("wgrad_x6_kernel",), "read_dword", "write_dword"),
    "wgrad_f32": (("wgrad_dma_kernel", "wgrad_kernel"), "read_x4", "write_dword"),
    "splitk_finalize": (("splitk_finalize",), "read_x4", "write_x4"),
    "reduce_partials": (("reduce_partials_kernel",), "read_dword", "write_dword"),
}
CAL_BYTES = 256 << 20


def short(name):
    return re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name))


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    acc = defaultdict(lambda: [0, 0.0])
    for name, cname, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if cname == counter:
            a = acc[short(name)]
            a[0] += 1
            a[1] += val
    return acc


def main(fetch_db, write_db, out, cal_fetch=None, cal_write=None):
    factors = {"read_x4": 2.0, "read_dword": 2.0, "read_rows8": 2.0, "write_dword": 1.0, "write_x4": 1.0}
    calibrated = False
    cal_raw = {}
    if cal_fetch and cal_write:
        cf, cw = per_kernel(cal_fetch, "FETCH_SIZE"), per_kernel(cal_write, "WRITE_SIZE")
        for k in ("read_x4", "read_dword", "read_rows8"):
            hit = [v for n, v in cf.items() if n.startswith(k)]
            if hit and hit[0][1] > 0:
                kb = hit[0][1] / hit[0][0]
                factors[k] = CAL_BYTES / (kb * 1024.0)
                cal_raw[k] = kb
        for k in ("write_dword", "write_x4"):
            hit = [v for n, v in cw.items() if n.startswith(k)]
            if hit and hit[0][1] > 0:
                kb = hit[0][1] / hit[0][0]
                factors[k] = CAL_BYTES / (kb * 1024.0)
                cal_raw[k] = kb
        calibrated = True
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for fam, (keys, rpat, wpat) in FAMILIES.items():
        nf = sum(v[0] for n, v in f.items() if n.startswith(keys))
        sf = sum(v[1] for n, v in f.items() if n.startswith(keys))
        nw = sum(v[0] for n, v in w.items() if n.startswith(keys))
        sw = sum(v[1] for n, v in w.items() if n.startswith(keys))
        if not nf or not nw:
            continue
        fa, wa = sf / nf, sw / nw
        res[fam] = {"launches": nf, "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa,
                    "read_factor": factors[rpat], "write_factor": factors[wpat],
                    "hbm_read_bytes_per_launch": factors[rpat] * fa * 1024, "hbm_write_bytes_per_launch": factors[wpat] * wa * 1024,
                    "hbm_bytes_per_launch": (factors[rpat] * fa + factors[wpat] * wa) * 1024}
    res["_factors"] = {"calibrated": calibrated, "factors": factors, "calibration_counter_KB_per_256MiB": cal_raw}
    res["_provenance"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py "
                          "--steps 2 --warmup 1 --no-graph` (v2, batch 32 x 65536, VAE phase); bytes = factor * counter KB * 1024 with "
                          "the factors " + ("measured in the same call on tools/probe/fetch_calib (256 MiB per access pattern)"
                                            if calibrated else "of the guide (wide streaming reads: FETCH x 2, uncalibrated for these patterns)"))
    # stamp: the library the counters were collected on -- bench.py reports `traffic` only while it loads the same one
    import hashlib, os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rave_amd", "librave_hip.so")
    res["librave_hip_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "_provenance"}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
