mkdir -p gpurun_out/r4x
{
for rep in 1 2; do
for e in 1 0; do
  echo -n "exact=$e "; ASIF_HIP_QP_INV_EXACT=$e python tools/dev_rz_time.py 100Hz 8 2>/dev/null | tail -1
done
done
} > gpurun_out/r4x/ab_rz38.txt 2>&1
