"""Kernels of ONE replayed hipGraph step: the difference of two rocprofv3 kernel traces of `bench.py` that differ only in
the number of timed steps (warm-up, capture and set-up cancel), divided by the difference in steps.

    python tools/prof_diff.py short.db long.db N_STEPS_DIFFERENCE [out.md]
"""
import re, sqlite3, sys


def table(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    return {n: (c, s) for n, c, s in cur.execute(f"select {name_col}, count(*), sum(end-start) from kernels group by {name_col}")}


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"void ", "", n)[:110]


def main(a, b, nsteps, out=None):
    ta, tb = table(a), table(b)
    rows = []
    for n, (c, s) in tb.items():
        c0, s0 = ta.get(n, (0, 0))
        if c - c0 > 0:
            rows.append((short(n), (c - c0) / nsteps, (s - s0) / 1e6 / nsteps))
    rows.sort(key=lambda r: -r[2])
    lines = ["| kernel | dispatches per step | ms per step | avg us |", "|---|---|---|---|"]
    for n, c, ms in rows:
        lines.append(f"| {n} | {c:.1f} | {ms:.3f} | {1e3 * ms / c:.1f} |")
    lines.append(f"\nrows listed: {sum(r[1] for r in rows):.1f} dispatches, {sum(r[2] for r in rows):.3f} ms of kernel time per replayed step "
                 f"(both streams added up)")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
