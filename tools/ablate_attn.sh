#!/bin/bash
# Build libhftt_ablate.so (attn_fwd.hip / attn_bwd.hip with -DHFTT_ATTN_ABLATE, the other objects as built) and time the attention kernels with
# single mechanisms switched off (see the ABL comments in the two sources).  Run on the GPU box AFTER nylon-amt_amd/build.py.
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
for f in attn_fwd attn_bwd; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_ATTN_ABLATE -x hip -c csrc/$f.hip -o build/${f}_ablate.o &
done
wait
OBJS=$(ls build/*.o | grep -v "attn_fwd\|attn_bwd\|_ablate")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_ablate.so $OBJS build/attn_fwd_ablate.o build/attn_bwd_ablate.o
cd ..
for bits in ${ABLATE_BITS:-0 1 2 4 8 16 32 12 18 63}; do
  echo "### HFTT_ATTN_ABLATE=$bits"
  HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_ablate.so HFTT_ATTN_ABLATE=$bits python tools/bench_attn.py ${ABLATE_WHAT:-both}
done
