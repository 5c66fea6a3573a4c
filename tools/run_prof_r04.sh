#!/bin/bash
# Round 4, GPU box: the rocprofv3 passes and bench records of the round, in parts that fit one gpurun call each.
#   tools/run_prof_r04.sh a|b|c|d      (then, on the host: tools/collect_r04.sh)
# a: the BASELINE configs -- direct-launch traces (tools/prof_direct.sh: the trace roofline.frac_rocprof is recomputed
#    from) and the five passes of tools/prof_cfg.sh for the default line, C2, C3, C4, C5
# b: the other filter configs (C6..C12) and C2 at 16 M instances
# c: bench records (driver-style default line, every config, batch sweeps, the --gpus 4 rehearsal on one GPU)
# d: the pre-assembled QP shapes
O=gpurun_out/final_r04; mkdir -p $O
case "$1" in
a)
  for c in 2 3 4 5; do bash tools/prof_direct.sh c${c}_direct --config $c > /dev/null 2>&1; echo "direct c$c done"; done
  bash tools/prof_cfg.sh default > /dev/null 2>&1; echo "prof default done"
  for c in 2 3 4 5; do bash tools/prof_cfg.sh c$c --config $c > /dev/null 2>&1; echo "prof c$c done"; done
  ;;
b)
  for c in 6 7 8 9 10 11 12; do bash tools/prof_cfg.sh c$c --config $c > /dev/null 2>&1; echo "prof c$c done"; done
  bash tools/prof_cfg.sh c2_batch16m --config 2 --batch 16777216 --steps 20 --warmup 3 > /dev/null 2>&1; echo "prof 16m done"
  ;;
c)
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/default_driver_style_bench.json 2> $O/default.err; echo "driver-style rc $?"
  timeout -k 10 300 python bench.py > $O/default_bench.json 2>> $O/default.err; echo "default rc $?"
  for c in 2 3 4 5 6 7 8 9 10 11 12; do timeout -k 10 300 python bench.py --config $c > $O/c${c}_bench.json 2> $O/c${c}_bench.err; echo "config $c rc $?"; done
  # the N > 1 path of bench.py on one GPU: four ranks on cuda:0 (the box allows six processes on the card; the 8-rank
  # shard arithmetic is tests/test_sharding_gloo.py, the 8-way split on one device tests/test_gpu_multi.py)
  ASIF_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 4 --no-cpu-baseline --no-pcie > $O/rehearsal_gpus4_on_one_gpu.json 2> $O/rehearsal.err; echo "rehearsal rc $?"
  for b in 262144 1048576 4194304 16777216; do timeout -k 10 300 python bench.py --config 2 --batch $b --steps 50 --graph 1 --no-cpu-baseline --no-pcie > $O/c2_batch$b.json 2>/dev/null; done
  timeout -k 10 300 python bench.py --config 3 --batch 65536 --steps 20 --no-cpu-baseline --no-pcie > $O/c3_batch65536.json 2>/dev/null
  timeout -k 10 300 python bench.py --config 4 --batch 131072 --steps 20 --no-cpu-baseline --no-pcie > $O/c4_batch131072_fused_pass.json 2>/dev/null
  echo "records done"
  ;;
d)
  for sh in c2 c3 c4 c5full; do bash tools/prof_cfg.sh qp_$sh --config qp --shape $sh > /dev/null 2>&1; echo "prof qp $sh done"; done
  for sh in c2 c3 c4 c5full; do timeout -k 10 300 python bench.py --config qp --shape $sh > $O/qp_${sh}_bench.json 2>/dev/null; done
  timeout -k 10 300 python tools/dev_rz_time.py 100Hz 8 > $O/rz38_time.txt 2>/dev/null
  echo "qp done"
  ;;
esac
