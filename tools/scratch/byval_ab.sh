for r in 1 2 3; do for l in asif_amd/libasif_hip.so asif_amd/csrc/build/variants/libasif_byval.so; do for k in 200 20; do
  ASIF_HIP_LIB=$PWD/$l timeout -k 10 200 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l'.split('/')[-1], 'steps', d['steps'], round(d['ms_per_step']*1e3,3))"
done; done; done
