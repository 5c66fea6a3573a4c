#!/bin/bash
# The trace the headline roofline is recomputed from: ONE rocprofv3 kernel-trace pass of the
# direct-launch leg only -- no graph replay, no PCIe leg, no CPU baseline, nothing else in the process -- so that
# the per-kernel AverageNs of the committed CSV is the average duration of the launches `value` is quoted on.
#   tools/prof_direct.sh <tag> <bench args...>     e.g.  tools/prof_direct.sh c2_direct --config 2
# -> gpurun_out/prof_<tag>/trace/t_results.db ; host: tools/summarize_prof.py --direct r04 c2_direct
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-c1 --graph 0 \
    --steps 200 --warmup 20 "$@" > $OUT/trace.log 2>&1 || echo "trace pass failed"
echo "done $TAG"
