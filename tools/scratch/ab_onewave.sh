mkdir -p gpurun_out/r4x
{
for rep in 1 2; do
for cfg in "product:" "onewave:ASIF_HIP_LIB=$PWD/asif_amd/csrc/build/ab/one_wave.so"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for b in 16384 32768 65536; do
    echo -n "$name "; env $envs python bench.py --config qp --shape c5full --batch $b --no-cpu-baseline --no-pcie --steps 30 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('18x12 per $b us', round(d['roofline']['kernel_avg_us'],1))"
    echo -n "$name "; env $envs python tools/scratch/bench_rd22.py $b 2>/dev/null | tail -1
  done
done
done
} > gpurun_out/r4x/ab_onewave.txt 2>&1
