#!/bin/bash
# one rocprofv3 --pmc pass (counters only, no trace domain) of a bench command; prints mean per launch of each counter
# for the kernels whose name matches <pattern>:   tools/pmc_once.sh <outdir> <pattern> "<counters>" <bench args...>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; PAT=$2; CNT=$3; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc -- python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 5 --warmup 1 "$@" > $OUT/pmc.log 2>&1 || echo "pmc pass failed"
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections, os
files = glob.glob(os.path.join(sys.argv[1], "pmc", "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY
