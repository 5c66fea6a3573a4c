(tools/dev_inv_ab.sh asif_amd/libasif_hip.so) > gpurun_out/inv_unroll.txt 2>&1
cat gpurun_out/inv_unroll.txt
timeout -k 10 900 python -m pytest tests/test_gpu_qp_lds.py tests/test_gpu_qp_generic.py -x -q 2>&1 | tail -3
