#!/bin/bash
# rocprofv3 passes for one bench config (run on the GPU box through gpurun): kernel trace + stats, then the two
# HBM counters in their own runs (counters are never combined with a trace domain).
#   tools/prof_cfg.sh <tag> <config> [extra bench args]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; CFG=$2; shift 2
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py --config $CFG --steps 200 --warmup 20 --no-cpu-baseline --no-pcie "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-pcie "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-pcie "$@" > $OUT/pmc_write.log 2>&1
echo "done $TAG"
