#!/bin/bash
# A developer build of the library with extra -D flags for ONE translation unit, next to the product build:
#   tools/build_variant.sh <name> <file.hip> "<flags>"   ->  asif_amd/csrc/build/ab/<name>.so   (git-ignored, travels with gpurun)
# Run it on the GPU box with ASIF_HIP_LIB=$PWD/asif_amd/csrc/build/ab/<name>.so (tools/ab_lib.sh, tools/ab_outputs.py).
set -e
cd "$(dirname "$0")/../asif_amd/csrc"
N=$1; F=$2; FLAGS=$3
mkdir -p build/ab
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function \
    -mllvm -amdgpu-sched-strategy=max-ilp $FLAGS -c $F -o build/ab/$N.o
OBJS=$(ls build/*.o | grep -v "build/${F%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/$N.so $OBJS build/ab/$N.o
echo built build/ab/$N.so
