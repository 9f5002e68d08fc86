#!/bin/bash
# GPU box: build libhftt_abl2.so (strip_gemm2.hip with -DHFTT_STRIP2_ABLATE, the other objects as built by nylon-amt_amd/build.py) and time the
# bf16 fused feed-forward block with single mechanisms switched off (HFTT_STRIP2_DEBUG bits: csrc/strip_gemm2.hip), one and two workgroups per CU.
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_STRIP2_ABLATE -x hip -c csrc/strip_gemm2.hip -o build/strip_gemm2_ablate.o
OBJS=$(ls build/*.o | grep -v "strip_gemm2\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_abl2.so $OBJS build/strip_gemm2_ablate.o
cd ..
for wpc in 1 2; do
  for bits in ${ABLATE_BITS:-0 1 2 3 64 128 192 256 259 451 512}; do
    echo "### WPC=$wpc HFTT_STRIP2_DEBUG=$bits"
    HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_abl2.so HFTT_MLP2_WPC=$wpc HFTT_STRIP2_DEBUG=$bits python tools/bench_strip.py ffn
  done
done
