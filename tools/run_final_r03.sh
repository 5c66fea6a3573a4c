# Round 3, GPU box: every rocprofv3 pass and bench record of the round in one call.
#   tools/run_final_r03.sh      (then, on the host: tools/collect_r03.sh)
O=gpurun_out/final_r03; mkdir -p $O
# profile passes (kernel trace + FETCH / WRITE / SQ counter passes, each its own run): tools/prof_cfg.sh
bash tools/prof_cfg.sh default > /dev/null 2>&1           # the driver's command: C2 + configs C3, C4, C5 + C1
for c in 2 3 4 5 6 7 8 9 10 11 12; do bash tools/prof_cfg.sh c$c --config $c > /dev/null 2>&1; echo "prof c$c done"; done
for sh in c2 c3 c4 c5full; do bash tools/prof_cfg.sh qp_$sh --config qp --shape $sh > /dev/null 2>&1; done
bash tools/prof_cfg.sh qp_c5full_wave_polish0 --config qp --shape c5full --lanes 64 --polish 0 > /dev/null 2>&1
bash tools/prof_cfg.sh c2_batch16m --config 2 --batch 16777216 --steps 20 --warmup 3 > /dev/null 2>&1
echo "profiles done"
# bench records
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/default_driver_style_bench.json 2> $O/default.err
timeout -k 10 300 python bench.py > $O/default_bench.json 2>> $O/default.err
for c in 2 3 4 5 6 7 8 9 10 11 12; do timeout -k 10 300 python bench.py --config $c > $O/c${c}_bench.json 2> $O/c${c}_bench.err; echo "config $c rc $?"; done
for p in 0 1; do timeout -k 10 300 python bench.py --config 2 --polish $p > $O/c2polish${p}_bench.json 2>/dev/null; done
for sh in c2 c3 c4 c5full; do timeout -k 10 300 python bench.py --config qp --shape $sh > $O/qp_${sh}_bench.json 2>/dev/null; done
timeout -k 10 300 python bench.py --config qp --shape c5full --lanes 64 --polish 0 > $O/qp_c5full_wave_polish0_bench.json 2>/dev/null
ASIF_HIP_QP_INV=0 timeout -k 10 300 python bench.py --config qp --shape c5full > $O/qp_c5full_wave_per_qp_bench.json 2>/dev/null
ASIF_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --no-pcie > $O/rehearsal_gpus2_on_one_gpu.json 2> $O/rehearsal.err
for b in 262144 1048576 4194304 16777216; do timeout -k 10 300 python bench.py --config 2 --batch $b --steps 50 --graph 1 --no-cpu-baseline --no-pcie > $O/c2_batch$b.json 2>/dev/null; done
timeout -k 10 300 python bench.py --config 2 --batch 16777216 --steps 50 --graph 1 --state-scale 0.5 --no-cpu-baseline --no-pcie > $O/c2_batch16777216_all_feasible.json 2>/dev/null
timeout -k 10 300 python bench.py --config 3 --batch 65536 --steps 20 --no-cpu-baseline --no-pcie > $O/c3_batch65536.json 2>/dev/null
timeout -k 10 300 python tools/dev_rz_time.py 100Hz 8 > $O/rz38_time.txt 2>/dev/null
tools/scratch/stream_pattern > $O/stream_pattern.txt 2>/dev/null
echo done
