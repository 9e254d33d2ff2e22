"""Assemble profiles/round1_pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; .db files):
HBM bytes per launch of the kernel families bench.py prices, bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950:
FETCH_SIZE counts half of a wide coalesced read -- calibrated on a 256 MiB copy; see profiles/README.md)."""
import json, re, sqlite3, sys
from collections import defaultdict

FAMILIES = {"pqmf": ("pqmf_",), "conv_igemm(fwd+dgrad)": ("conv_x6_kernel", "conv_igemm_dma_kernel", "conv_igemm_kernel"),
            "conv_wgrad": ("wgrad_dma_kernel", "wgrad_kernel", "wgrad_x6_kernel")}


def per_family(db, counter):
    cur = sqlite3.connect(db).cursor()
    acc = defaultdict(lambda: [0, 0.0])
    for name, cname, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if cname != counter:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name))
        for fam, keys in FAMILIES.items():
            if short.startswith(keys):
                acc[fam][0] += 1
                acc[fam][1] += val
    return acc


def main(fetch_db, write_db, out):
    f, w = per_family(fetch_db, "FETCH_SIZE"), per_family(write_db, "WRITE_SIZE")
    res = {}
    for fam in FAMILIES:
        nf, sf = f[fam]
        nw, sw = w[fam]
        if not nf or not nw:
            continue
        fa, wa = sf / nf, sw / nw
        res[fam] = {"launches": nf, "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa, "hbm_bytes_per_launch": (2 * fa + wa) * 1024}
    res["_provenance"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py "
                          "--steps 2 --warmup 1` (v2, batch 32 x 65536, VAE phase), round 2, tools/final_measure.sh; bytes = "
                          "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts half of a wide coalesced read (calibrated on a "
                          "256 MiB copy: FETCH 128 MiB, WRITE 256 MiB); the factor for the LDS-DMA reads of these kernels is "
                          "uncalibrated, so the read side is an upper-bound estimate")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "_provenance"}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
