for r in 1 2 3; do for k in 200 20; do
  timeout -k 10 200 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps', d['steps'], round(d['ms_per_step']*1e3,3), d['parity'] if 'parity' in d else '')"
done; done
