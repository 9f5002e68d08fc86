#!/bin/bash
# Build libhftt_stamps.so (strip_gemm2.hip with -DHFTT_STRIP_STAMPS, the other objects as built) and print the per-slot phase times of the
# pipelined fused feed-forward block (tools/stamp_mlp2.py).  Run on the GPU box AFTER nylon-amt_amd/build.py.
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_STRIP_STAMPS -x hip -c csrc/strip_gemm2.hip -o build/strip_gemm2_stamps.o
OBJS=$(ls build/*.o | grep -v "strip_gemm2\|strip_gemm[345]\|\.x\.o\|\.g\.o\|_ablate\|_stamps\|_g8")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_stamps.so $OBJS build/strip_gemm2_stamps.o
cd ..
for e in ${EXTRAS:-0}; do echo "### extra debug bits $e"; EXTRA=$e HFTT_MLP2_PATCH=0 HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_stamps.so python tools/stamp_mlp2.py; done
