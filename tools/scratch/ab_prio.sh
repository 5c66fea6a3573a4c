mkdir -p gpurun_out/r4x
{
for rep in 1 2; do
for cfg in "product:" "prio1:ASIF_HIP_LIB=$PWD/asif_amd/csrc/build/ab/prio1.so" "prio2:ASIF_HIP_LIB=$PWD/asif_amd/csrc/build/ab/prio2.so"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for b in 8192 32768; do
    echo -n "$name "; env $envs python bench.py --config qp --shape c5full --batch $b --no-cpu-baseline --no-pcie --steps 50 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('18x12 per $b us', round(d['roofline']['kernel_avg_us'],1))"
  done
  echo -n "$name "; env $envs python tools/scratch/bench_rd22.py 8192 2>/dev/null | tail -1
done
done
} > gpurun_out/r4x/ab_prio.txt 2>&1
