mkdir -p gpurun_out/r4x
for e in 1 0 1 0; do
  for b in 8192 512 32768; do
    ASIF_HIP_QP_INV_EXACT=$e python bench.py --config qp --shape c5full --batch $b --no-cpu-baseline --no-pcie --steps 50 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('exact=$e 18x12 per $b us', round(d['roofline']['kernel_avg_us'],1), 'parity', d.get('parity'))"
  done
done > gpurun_out/r4x/ab_exact.txt 2>&1
