#!/bin/bash
# GPU box, after nylon-amt_amd/build.py: the issue pattern of the interleaved dQ steps (HFTT_XB_SG_A / _B of csrc/x3_attn_bwd.h) swept on the encoder shape
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
OBJS=$(ls build/*.o | grep -v "x3_attn_pl\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate\|_stamps\|_nocap\|_sg")
for ab in ${SWEEP:-"10 6" "4 4" "16 4" "20 8" "8 12" "0 0" "28 0"}; do
  set -- $ab
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_XB_SG_A=$1 -DHFTT_XB_SG_B=$2 -x hip -c csrc/x3_attn_pl.hip -o build/x3_attn_pl_sg.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_sg.so $OBJS build/x3_attn_pl_sg.o
  echo -n "A=$1 B=$2: "
  HFTT_LIB_PATH=$PWD/lib/libhftt_sg.so NSEQ=1024 LQ=256 LK=256 python ../tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
done
