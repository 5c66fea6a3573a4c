mkdir -p gpurun_out/r4x
{
for rep in 1 2; do
for cfg in "exact1:" "exact0:ASIF_HIP_QP_INV_EXACT=0" "twowaves:ASIF_HIP_LIB=$PWD/asif_amd/csrc/build/ab/two_waves.so" "twowaves_pad:ASIF_HIP_QP_INV_EXACT=0 ASIF_HIP_LIB=$PWD/asif_amd/csrc/build/ab/two_waves.so"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for b in 8192 512 2048; do
    echo -n "$name "; env $envs python tools/scratch/bench_rd22.py $b 2>/dev/null | tail -1
  done
  for b in 8192 512; do
    echo -n "$name "; env $envs python bench.py --config qp --shape c5full --batch $b --no-cpu-baseline --no-pcie --steps 50 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('18x12 per $b us', round(d['roofline']['kernel_avg_us'],1))"
  done
done
done
} > gpurun_out/r4x/ab_rd22.txt 2>&1
