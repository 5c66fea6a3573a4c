# Round 4, GPU box: the bench records again once the counter summaries they cite are in profiles/r04 (a record quotes
# `traffic` and `roofline.valu` from the committed summary of its own command: tools/run_prof_r04.sh, then
# tools/collect_r04.sh on the host, then this).   tools/bench_records_r04.sh [cfg ...]   default: every config
O=gpurun_out/final_r04; mkdir -p $O
CFGS="$@"; [ -z "$CFGS" ] && CFGS="2 3 4 5 6 7 8 9 10 11 12"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/default_driver_style_bench.json 2> $O/default.err; echo "driver-style rc $?"
timeout -k 10 300 python bench.py > $O/default_bench.json 2>> $O/default.err; echo "default rc $?"
for c in $CFGS; do timeout -k 10 300 python bench.py --config $c > $O/c${c}_bench.json 2> $O/c${c}_bench.err; echo "config $c rc $?"; done
