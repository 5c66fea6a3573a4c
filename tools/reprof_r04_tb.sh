bash tools/prof_direct.sh c4_direct --config 4 > /dev/null 2>&1; echo "direct c4 done"
bash tools/prof_cfg.sh default > /dev/null 2>&1; echo "prof default done"
for c in 4 8 12; do bash tools/prof_cfg.sh c$c --config $c > /dev/null 2>&1; echo "prof c$c done"; done
O=gpurun_out/final_r04; mkdir -p $O
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/default_driver_style_bench.json 2> $O/default.err; echo "driver-style rc $?"
timeout -k 10 300 python bench.py > $O/default_bench.json 2>> $O/default.err; echo "default rc $?"
for c in 4 8 12; do timeout -k 10 300 python bench.py --config $c > $O/c${c}_bench.json 2> $O/c${c}_bench.err; echo "config $c rc $?"; done
ASIF_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 4 --no-cpu-baseline --no-pcie > $O/rehearsal_gpus4_on_one_gpu.json 2> $O/rehearsal.err; echo "rehearsal rc $?"
