#!/bin/bash
# GPU box: build libhftt_xattn.so (x3_attn.hip with -DHFTT_X3_ATTN_ABLATE, the other objects as built by nylon-amt_amd/build.py) and time the
# encoder-shaped x3 attention backward with single mechanisms switched off (bits: csrc/x3_attn.hip).
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_X3_ATTN_ABLATE -x hip -c csrc/x3_attn.hip -o build/x3_attn_ablate.o
OBJS=$(ls build/*.o | grep -v "/x3_attn\.o\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_xattn.so $OBJS build/x3_attn_ablate.o
cd ..
for bits in ${ABLATE_BITS:-0 1 2 4 8 16 32 63}; do
  HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_xattn.so HFTT_X3_ATTN_DEBUG=$bits CROSS=0 python tools/bench_x3_attn.py 2>/dev/null | grep "^debug"
done
