#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ns) of a rocprofv3 --kernel-trace results .db, as CSV.
    python tools/rocprof_db_stats.py gpurun_out/prof/x_results.db > profiles/rNN/name_kernel_stats.csv
    python tools/rocprof_db_stats.py x_results.db --window 20:220 --json     the dispatches [20, 220) of each asif kernel
in dispatch order (the timed region of `bench.py --warmup 20 --steps 200`): the same columns plus the span from the
first start to the last end divided by the count (back-to-back launch interval, what an event pair around the region
divides out) as one JSON object."""
import json
import sqlite3
import sys

args = sys.argv[1:]
db = sqlite3.connect(args[0])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
if "--window" in args:
    lo, hi = (int(v) for v in args[args.index("--window") + 1].split(":"))
    per = {}
    for name, start, end in cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on "
                                        f"d.kernel_id=s.id order by d.start"):
        per.setdefault(name, []).append((start, end))
    out = {}
    for name, v in per.items():
        if "asif" not in name:
            continue
        w = v[lo:hi]
        if not w:
            continue
        dur = [e - s for s, e in w]
        out[name] = {"dispatches_total": len(v), "window": [lo, min(hi, len(v))], "calls": len(w),
                     "average_ns": sum(dur) / len(dur), "min_ns": min(dur), "max_ns": max(dur),
                     "median_ns": sorted(dur)[len(dur) // 2],
                     "span_ns_per_call": (w[-1][1] - w[0][0]) / len(w),
                     "overlapped_with_previous": sum(1 for a, b in zip(w, w[1:]) if b[0] < a[1])}
    print(json.dumps(out, indent=1))
    sys.exit(0)
rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                        f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for r in rows:
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (r[0], r[1], r[2], r[3], 100 * r[2] / tot, r[4], r[5]))
