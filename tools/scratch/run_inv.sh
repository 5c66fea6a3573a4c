(tools/dev_inv_ab.sh asif_amd/libasif_hip.so; python tools/dev_rz_time.py 10Hz_50pt 8 2>/dev/null | tail -1; ASIF_HIP_LIB=$PWD/asif_amd/csrc/build/ab/invprof.so python tools/dev_inv_sections.py 2>&1 | tail -13) > gpurun_out/inv_fewbp.txt 2>&1
cat gpurun_out/inv_fewbp.txt
timeout -k 10 900 python -m pytest tests/test_gpu_qp_lds.py tests/test_gpu_qp_generic.py tests/test_gpu_realizable.py tests/test_gpu_host_cpp.py -x -q 2>&1 | tail -3
