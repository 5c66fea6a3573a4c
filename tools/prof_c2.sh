#!/bin/bash
# rocprofv3 passes for the headline bench (run on the GPU box through gpurun):
#   1. kernel trace + stats  2. PMC FETCH_SIZE  3. PMC WRITE_SIZE   (counters in their own runs)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
CFG=${2:-2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 50 --warmup 10 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
