#!/bin/bash
# GPU box: build libhftt_xattn.so (x3_attn_pl.hip with -DHFTT_X3_ATTN_ABLATE, the other objects as built by nylon-amt_amd/build.py) and time the
# plane-operand x3 attention forward with single mechanisms switched off (HFTT_X3P_DEBUG bits: csrc/x3_attn_pl.hip).
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_X3_ATTN_ABLATE -x hip -c csrc/x3_attn_pl.hip -o build/x3_attn_pl_ablate.o
OBJS=$(ls build/*.o | grep -v "x3_attn_pl\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_xattn.so $OBJS build/x3_attn_pl_ablate.o
cd ..
for bits in ${ABLATE_BITS:-0 1 2 3 4 8 16 12 31}; do
  echo "HFTT_X3P_DEBUG=$bits"
  HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_xattn.so HFTT_X3P_DEBUG=$bits CROSS=0 python tools/bench_x3_attn.py 2>/dev/null | grep planes
done
