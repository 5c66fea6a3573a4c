#!/bin/bash
# A developer build of the library with extra -D flags for some translation units, next to the product build:
#   tools/build_variant.sh <name> "<file.hip> [<file.hip> ...]" "<flags>"   ->  asif_amd/csrc/build/ab/<name>.so
# (git-ignored, travels with gpurun).  Run it on the GPU box with ASIF_HIP_LIB=$PWD/asif_amd/csrc/build/ab/<name>.so
# (tools/ab_lib.sh, tools/ab_many.sh, tools/ab_outputs.py).
set -e
cd "$(dirname "$0")/../asif_amd/csrc"
N=$1; FILES=$2; FLAGS=$3
mkdir -p build/ab
make -s -j8 >/dev/null
OBJS=$(ls build/*.o)
for F in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function \
      -mllvm -amdgpu-sched-strategy=max-ilp $FLAGS -c $F -o build/ab/$N.${F%.hip}.o
  OBJS=$(echo "$OBJS" | grep -v "build/${F%.hip}.o")
  OBJS="$OBJS build/ab/$N.${F%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/$N.so $OBJS
echo built build/ab/$N.so
