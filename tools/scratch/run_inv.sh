(for KJ in 1 0; do echo "== keep_kj $KJ"; export ASIF_HIP_QP_KEEP_KJ=$KJ; python tools/dev_rz_time.py 10Hz 2 2>/dev/null | tail -1; ASIF_HIP_QP_INV=0 python tools/dev_rz_time.py 10Hz_50pt 4 2>/dev/null | tail -1; ASIF_HIP_QP_INV=0 python tools/dev_rz_time.py 100Hz 8 2>/dev/null | tail -1;
 ASIF_HIP_QP_INV=0 python bench.py --config qp --shape c5full --no-cpu-baseline --no-pcie --steps 30 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('18x12 on qp_lds per 8192 us', round(d['roofline']['kernel_avg_us'],1))"; done) > gpurun_out/lds_keepkj.txt 2>&1
cat gpurun_out/lds_keepkj.txt
unset ASIF_HIP_QP_KEEP_KJ
timeout -k 10 900 python -m pytest tests/test_gpu_qp_lds.py tests/test_gpu_qp_generic.py tests/test_gpu_realizable.py tests/test_gpu_host_cpp.py -x -q 2>&1 | tail -5
ASIF_HIP_QP_INV=0 timeout -k 10 900 python -m pytest tests/test_gpu_qp_lds.py tests/test_gpu_qp_generic.py -x -q 2>&1 | tail -3
