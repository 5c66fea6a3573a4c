for g in 0 1; do
  for k in 300 20; do
    timeout -k 10 100 python bench.py --steps $k --warmup 5 --graph $g --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', $g, 'steps', $k, d['ms_per_step'])"
  done
done
