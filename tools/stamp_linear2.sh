#!/bin/bash
# Build libhftt_stamps.so (strip_gemm2.hip with -DHFTT_STRIP_STAMPS, the other objects as built) and print the per-slot phase times of the
# pipelined strip linear kernel (tools/stamp_linear2.py).  Run on the GPU box AFTER nylon-amt_amd/build.py.
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
for f in strip_gemm2 strip_gemm3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_STRIP_STAMPS -x hip -c csrc/$f.hip -o build/${f}_stamps.o &
done
wait
OBJS=$(ls build/*.o | grep -v "strip_gemm2\|strip_gemm3\|_ablate\|_stamps")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_stamps.so $OBJS build/strip_gemm2_stamps.o build/strip_gemm3_stamps.o
cd ..
HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_stamps.so python tools/stamp_linear2.py
