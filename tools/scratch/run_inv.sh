(for L in asif_amd/csrc/build/ab/inv_w2all.so; do tools/dev_inv_ab.sh $L; ASIF_HIP_LIB=$PWD/$L python tools/dev_inv_occupancy.py 2>&1 | tail -14; done) > gpurun_out/inv_w2all.txt 2>&1
cat gpurun_out/inv_w2all.txt
