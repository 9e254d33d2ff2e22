"""Timeline of ONE replayed hipGraph step from a rocprofv3 kernel trace (.db) of `bench.py`: per-queue busy time, the union of
busy intervals, the idle time between kernels, and the dispatch list of the step (start offset, duration, queue, kernel).
A step is delimited by consecutive prep_scales_kernel dispatches (the first kernel of every step).

    python tools/prof_timeline.py trace.db [out.txt]
"""
import re, sqlite3, sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = [c for c in cols if "queue" in c.lower() or "stream" in c.lower()]
    qcol = qcol[0] if qcol else None
    rows = cur.execute(f"select {name_col}, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "prep_scales_kernel" in r[0]]
    if len(marks) < 12:
        print("not enough steps in the trace"); return
    a, b = marks[-6], marks[-5]          # one steady-state replayed step
    step = rows[a:b]
    t0 = step[0][1]
    wall = rows[b][1] - t0
    lines = [f"step wall (first kernel to the next step's first kernel): {wall / 1e3:.1f} us, {len(step)} dispatches"]
    ivs = sorted((r[1], r[2]) for r in step)
    busy, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
    gaps = []
    for s, e in ivs[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    lines.append(f"union of busy intervals {busy / 1e3:.1f} us; idle (no kernel on any queue) {sum(g for g, _ in gaps) / 1e3:.1f} us in {len(gaps)} gaps; "
                 f"sum of kernel durations {sum(e - s for s, e in ivs) / 1e3:.1f} us")
    gaps.sort(reverse=True)
    lines.append("largest gaps (us @ offset us): " + ", ".join(f"{g / 1e3:.1f}@{o / 1e3:.0f}" for g, o in gaps[:12]))
    if qcol:
        qs = {}
        for r in step:
            qs.setdefault(r[3], [0, 0])
            qs[r[3]][0] += r[2] - r[1]; qs[r[3]][1] += 1
        lines.append("per queue: " + ", ".join(f"{q}: {v[0] / 1e3:.0f} us in {v[1]} dispatches" for q, v in qs.items()))
    lines.append("")
    lines.append("offset_us  dur_us  queue  kernel")
    for r in step:
        n = re.sub(r"\(anonymous namespace\)::|void ", "", r[0])[:90]
        lines.append(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f}  {r[3] if qcol else '-'}  {n}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print("\n".join(lines[:6]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
