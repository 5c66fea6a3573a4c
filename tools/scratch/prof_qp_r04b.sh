# re-profile of the lifted 18 x 12 shape after the exact-size instantiation, and the realizable shapes' timings
O=gpurun_out/final_r04; mkdir -p $O
bash tools/prof_cfg.sh qp_c5full --config qp --shape c5full > /dev/null 2>&1; echo "prof qp c5full done"
timeout -k 10 300 python bench.py --config qp --shape c5full > $O/qp_c5full_bench.json 2>/dev/null
for k in 100Hz 10Hz_50pt; do timeout -k 10 300 python tools/dev_rz_time.py $k 8 2>/dev/null | tail -1; done > $O/rz38_time.txt
for b in 512 2048 8192 16384 65536; do python tools/scratch/bench_rd22.py $b 2>/dev/null | tail -1; done > $O/rd22_time.txt
for b in 512 32768 65536; do python bench.py --config qp --shape c5full --batch $b --no-cpu-baseline --no-pcie --steps 50 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('18x12 per $b: kernel us', round(d['roofline']['kernel_avg_us'],1))"; done > $O/qp_c5full_batches.txt
echo done
