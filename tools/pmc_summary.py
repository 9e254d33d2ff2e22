"""Summarise rocprofv3 --pmc results (.db): per kernel (short name) average counter values."""
import re, sqlite3, sys
from collections import defaultdict

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]{0,60})?)", n)
    return (m.group(1) if m else n)[:70]

def main(path, filt=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for name, cname, val, d, did in cur.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection"):
        s = short(name)
        if filt and filt not in s:
            continue
        agg[s][cname].append(val)
        dur[s].append(d)
    for k in agg:
        print(f"== {k}  (dispatches {len(next(iter(agg[k].values())))}, avg duration {sum(dur[k])/len(dur[k])/1e3:.1f} us)")
        for c, v in sorted(agg[k].items()):
            print(f"   {c:32s} avg {sum(v)/len(v):14.1f}")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
