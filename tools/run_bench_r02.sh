# bench records of the round (GPU box): one bench line per config into gpurun_out/final_r02, copied to profiles/r02 afterwards
O=gpurun_out/final_r02; mkdir -p $O
for c in 2 3 4 5 6 7 8 9 10 11; do timeout -k 10 300 python bench.py --config $c > $O/c${c}_bench.json 2> $O/c${c}_bench.err; echo "config $c rc $?"; done
timeout -k 10 100 python bench.py --steps 20 --warmup 5 > $O/c2_driver_style_bench.json 2>/dev/null
for p in 0 1; do timeout -k 10 300 python bench.py --config 2 --polish $p > $O/c2polish${p}_bench.json 2> $O/c2polish${p}_bench.err; done
for sh in c2 c3 c4 c5full; do timeout -k 10 300 python bench.py --config qp --shape $sh > $O/qp_${sh}_bench.json 2> $O/qp_${sh}_bench.err; done
timeout -k 10 300 python bench.py --config qp --shape c5full --lanes 64 --polish 0 > $O/qp_c5full_wave_polish0_bench.json 2> /dev/null
ASIF_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --no-pcie > $O/rehearsal_gpus2_on_one_gpu.json 2> $O/rehearsal.err
for b in 262144 1048576 4194304 16777216; do timeout -k 10 300 python bench.py --config 2 --batch $b --steps 50 --no-cpu-baseline --no-pcie > $O/c2_batch$b.json 2>/dev/null; done
timeout -k 10 300 python bench.py --config 3 --batch 65536 --steps 20 --no-cpu-baseline --no-pcie > $O/c3_batch65536.json 2>/dev/null
echo done
