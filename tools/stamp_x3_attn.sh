#!/bin/bash
# GPU box, after nylon-amt_amd/build.py: build x3_attn_pl.hip with -DHFTT_X3_ATTN_STAMPS (interleaved dQ on and off) and print the phase
# times of one query block of the 256-key backward (tools/stamp_x3_attn.py).
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
OBJS=$(ls build/*.o | grep -v "x3_attn_pl\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate\|_stamps")
for il in ${ILS:-1 0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_X3_ATTN_STAMPS -DHFTT_XB_IL=$il $XDEFS -x hip -c csrc/x3_attn_pl.hip -o build/x3_attn_pl_stamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_stamps.so $OBJS build/x3_attn_pl_stamps.o
  echo "### HFTT_XB_IL=$il (units: s_memtime ticks)"
  HFTT_LIB_PATH=$PWD/lib/libhftt_stamps.so python ../tools/stamp_x3_attn.py 2>&1 | grep -v amdgpu.ids
done
