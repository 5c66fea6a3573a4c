(tools/dev_inv_ab.sh asif_amd/libasif_hip.so; python tools/dev_rz_time.py 10Hz_50pt 8 2>/dev/null | tail -1) > gpurun_out/inv_loop.txt 2>&1
cat gpurun_out/inv_loop.txt
timeout -k 10 900 python -m pytest tests/test_gpu_qp_lds.py tests/test_gpu_qp_generic.py tests/test_gpu_realizable.py -x -q 2>&1 | tail -5
