for r in 1 2; do for u in 0 1; do for k in 200 20; do
  BENCH_GRAPH_UPLOAD=$u timeout -k 10 200 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('upload', $u, 'steps', d['steps'], d['ms_per_step'], d['roofline']['kernel_avg_us'])"
done; done; done
