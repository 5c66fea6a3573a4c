(tools/dev_inv_ab.sh asif_amd/libasif_hip.so; python tools/dev_inv_occupancy.py 2>&1 | tail -14) > gpurun_out/inv_final.txt 2>&1
cat gpurun_out/inv_final.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
