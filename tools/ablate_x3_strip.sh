#!/bin/bash
# GPU box: build libhftt_xstrip.so (x3_strip.hip with -DHFTT_X3_STRIP_ABLATE, the other objects as built by nylon-amt_amd/build.py) and time the
# x3 strip kernels with single mechanisms switched off (HFTT_X3_DEBUG bits: csrc/x3_strip.hip).  QKV_ONLY=1 / FFN_ONLY=1 narrow the run.
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_X3_STRIP_ABLATE -x hip -c csrc/x3_strip.hip -o build/x3_strip_ablate.o
OBJS=$(ls build/*.o | grep -v "x3_strip\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_xstrip.so $OBJS build/x3_strip_ablate.o
cd ..
for bits in ${ABLATE_BITS:-0 16 32 48 1 2}; do
  HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_xstrip.so HFTT_X3_DEBUG=$bits python tools/bench_x3.py
done
