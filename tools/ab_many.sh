#!/bin/bash
# Timing of several builds of the library on the SAME box, two rounds, and bitwise outputs of each against the first:
#   tools/ab_many.sh <dir> <config> <libA.so> <libB.so> ...
D=gpurun_out/$1; C=$2; shift 2
mkdir -p $D
for rep in 1 2; do
  for L in "$@"; do
    ASIF_HIP_LIB=$PWD/$L timeout -k 10 200 python bench.py --config $C --no-pcie --no-cpu-baseline --steps 100 --warmup 20 2>>$D/err.txt \
      | python -c "import json,sys; d=json.load(sys.stdin); print('$C', '$L', 'rep$rep', '%.2f us' % d['roofline']['kernel_avg_us'])" | tee -a $D/times.txt
  done
done
A=$1; shift
for L in "$@"; do
  timeout -k 10 300 python tools/ab_outputs.py $A $L $C 2>&1 | grep config | tee -a $D/bits.txt
done
