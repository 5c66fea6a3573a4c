#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ns) of a rocprofv3 --kernel-trace results .db, as CSV.
    python tools/rocprof_db_stats.py gpurun_out/prof/x_results.db > profiles/rNN/name_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                        f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for r in rows:
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (r[0], r[1], r[2], r[3], 100 * r[2] / tot, r[4], r[5]))
