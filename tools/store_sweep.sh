#!/bin/bash
# explicit filter, masked vs whole-line stores across batch sizes (same box): ASIF_HIP_WHOLE_LINES_FROM moves the switch-over
D=gpurun_out/$1; mkdir -p $D
for B in 65536 262144 1048576 4194304 16777216; do
  for from in 1 1000000000; do
    ASIF_HIP_WHOLE_LINES_FROM=$from timeout -k 10 200 python bench.py --config 2 --batch $B --no-pcie --no-cpu-baseline --graph 1 --steps 100 --warmup 20 2>>$D/err.txt \
      | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('B', $B, 'whole-line stores' if $from==1 else 'masked stores', '%.2f us' % r['kernel_avg_us'], '%.0f GB/s' % r['achieved'], 'frac %.3f' % r['frac'])"
  done
done
