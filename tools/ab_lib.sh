#!/bin/bash
# A/B of two builds of the library on the SAME box (devices differ by up to 12 % on compute-bound kernels):
#   tools/ab_lib.sh <dir> <libA.so> <libB.so> <config> [<config> ...]     configs: 2..11 or qp:<shape>
D=gpurun_out/$1; A=$2; B=$3; shift 3
mkdir -p $D
for c in "$@"; do
  for rep in 1 2; do
    for L in $A $B; do
      if [[ $c == qp:* ]]; then args="--config qp --shape ${c#qp:}"; else args="--config $c"; fi
      ASIF_HIP_LIB=$PWD/$L timeout -k 10 200 python bench.py $args --no-pcie --no-cpu-baseline --steps 100 --warmup 20 2>>$D/err.txt \
        | python -c "import json,sys; d=json.load(sys.stdin); print('$c', '$L', 'rep$rep', '%.2f us' % d['roofline']['kernel_avg_us'])"
    done
  done
done
