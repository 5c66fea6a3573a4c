#!/bin/bash
# Developer probe: the half-wave kernel's Newton-step histogram, time per 8 192 and per 512 lifted 18 x 12 problems, and
# the 38 x 29 realizable problems' time, for several builds of the library on the same box.
#   tools/dev_inv_ab.sh lib1.so lib2.so ...
for L in "$@"; do
  echo "== $L"
  ASIF_HIP_LIB=$PWD/$L python tools/dev_inv_hist.py 2>/dev/null | sed -n '1p;2p;$p' | cut -c1-400
  ASIF_HIP_LIB=$PWD/$L python bench.py --config qp --shape c5full --no-cpu-baseline --no-pcie --steps 50 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('18x12 per 8192 us', round(d['roofline']['kernel_avg_us'],1), 'parity', d.get('parity'))"
  ASIF_HIP_LIB=$PWD/$L python bench.py --config qp --shape c5full --batch 512 --no-cpu-baseline --no-pcie --steps 50 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('18x12 per 512 us', round(d['roofline']['kernel_avg_us'],1))"
  ASIF_HIP_LIB=$PWD/$L python tools/dev_rz_time.py 100Hz 8 2>/dev/null | tail -1
done
