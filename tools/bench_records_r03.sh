O=gpurun_out/final_r03; mkdir -p $O
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/default_driver_style_bench.json 2> $O/default.err
timeout -k 10 300 python bench.py > $O/default_bench.json 2>> $O/default.err
for c in 2 3 4 5 6 7 8 9 10 11 12; do timeout -k 10 300 python bench.py --config $c > $O/c${c}_bench.json 2> $O/c${c}_bench.err; echo "config $c rc $?"; done
for sh in c2 c3 c4 c5full; do timeout -k 10 300 python bench.py --config qp --shape $sh > $O/qp_${sh}_bench.json 2>/dev/null; done
timeout -k 10 300 python bench.py --config qp --shape c5full --lanes 64 --polish 0 > $O/qp_c5full_wave_polish0_bench.json 2>/dev/null
echo done
