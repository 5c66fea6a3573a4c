set -o pipefail
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"; cut -c1-600 $O/bench_default.json
for sh in c2 c3 c4 c5full; do timeout -k 10 200 python bench.py --config qp --shape $sh --steps 50 --warmup 5 > $O/qp_$sh.json 2> $O/qp_$sh.err; echo "qp $sh rc $?"; python -c "import json; d=json.load(open('$O/qp_$sh.json')); print(d['ms_per_step'], d['value'], d['config']['kernel'], d['config']['solver'], d['config']['status_histogram'], d.get('parity'), d['roofline']['frac'])"; done
for sh in c2 c5full; do timeout -k 10 200 python bench.py --config qp --shape $sh --lanes 64 --polish 0 --steps 20 --warmup 3 --no-cpu-baseline > $O/qp_${sh}_admmwave.json 2> $O/qp_${sh}_admmwave.err; echo "qp $sh admm wave rc $?"; python -c "import json; d=json.load(open('$O/qp_${sh}_admmwave.json')); print(d['ms_per_step'], d['value'], d['config']['kernel'], d['config']['solver'], d['config']['status_histogram'])"; done
timeout -k 10 200 python bench.py --config qp --shape c2 --lanes 64 --steps 20 --warmup 3 --no-cpu-baseline > $O/qp_c2_lds.json 2> $O/qp_c2_lds.err; python -c "import json; d=json.load(open('$O/qp_c2_lds.json')); print('c2 on lds', d['ms_per_step'], d['value'], d['config']['solver'], d['config']['status_histogram'])"
ASIF_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline --no-pcie > $O/rehearsal_gpus2.json 2> $O/rehearsal_gpus2.err; echo "rehearsal rc $?"; cut -c1-300 $O/rehearsal_gpus2.json
rocprofv3 -L > $O/counters.txt 2>&1; grep -c . $O/counters.txt
