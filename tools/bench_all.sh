#!/bin/bash
# Every filter config and QP shape through bench.py, one JSON line each, into gpurun_out/<dir>/ (GPU box).
#   tools/bench_all.sh <dir> [extra bench args]
D=gpurun_out/$1; shift
mkdir -p $D
for c in 2 3 4 5 6 7 8 9 10 11; do
  timeout -k 10 300 python bench.py --config $c --no-pcie "$@" > $D/c${c}_bench.json 2>> $D/bench.err || echo "config $c failed"
done
for s in c2 c3 c4 c5full; do
  timeout -k 10 300 python bench.py --config qp --shape $s "$@" > $D/qp_${s}_bench.json 2>> $D/bench.err || echo "qp $s failed"
done
python - "$D" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_bench.json"))):
    try:
        d = json.load(open(f))
        p = d.get("parity") or {}
        print(f"{os.path.basename(f):24s} {d['ms_per_step']*1e3:10.2f} us  value {d['value']:.4g}  kernel {d['roofline']['kernel_avg_us']:.2f} us  "
              f"rcmis {p.get('rc_mismatches', p.get('status_mismatches'))} err {p.get('max_abs_u_err_vs_exact', p.get('max_abs_err_vs_exact'))}")
    except Exception as e:
        print(f, "unreadable", e)
PY
