#!/bin/bash
# rocprofv3 passes for one bench command (run on the GPU box through gpurun): kernel trace + stats, then the
# counters in their own runs (counters are never combined with a trace domain; FETCH_SIZE and WRITE_SIZE do not fit
# one pass; the SQ block has 8 slots).
#   tools/prof_cfg.sh <tag> <bench args...>        e.g.  tools/prof_cfg.sh c2 --config 2
#                                                        tools/prof_cfg.sh qp_c5full --config qp --shape c5full
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-pcie"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B --steps 200 --warmup 20 "$@" > $OUT/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 10 --warmup 2 "$@" > $OUT/pmc_fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 10 --warmup 2 "$@" > $OUT/pmc_write.log 2>&1 || echo "write pass failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $B --steps 10 --warmup 2 "$@" > $OUT/pmc_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT --output-format csv -d $OUT/pmc_sq2 -- $B --steps 10 --warmup 2 "$@" > $OUT/pmc_sq2.log 2>&1 || echo "sq2 pass failed"
echo "done $TAG"
