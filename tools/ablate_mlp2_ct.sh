#!/bin/bash
# GPU box: the bf16 fused feed-forward block with mechanisms removed AT COMPILE TIME (one library per mask of HFTT_MLP2_CT, csrc/strip_gemm2.hip:
# 1 no ring fills, 2 no slot barrier, 4 no next-block row prefetch, 8 no fragment reads, 16 no slot wait) -- no run-time switch left in the slot
# bodies, so what remains is scheduled as in the product.  Results of the masked builds are garbage; times are the point.
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
OBJS=$(ls build/*.o | grep -v "strip_gemm2\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate\|_stamps\|_ct")
for m in ${CT_MASKS:-0 1 2 4 8 16 19 27 31}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_MLP2_CT=$m -x hip -c csrc/strip_gemm2.hip -o build/strip_gemm2_ct$m.o &
done
wait
for m in ${CT_MASKS:-0 1 2 4 8 16 19 27 31}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_ct$m.so $OBJS build/strip_gemm2_ct$m.o
done
cd ..
for m in ${CT_MASKS:-0 1 2 4 8 16 19 27 31}; do
  echo "### HFTT_MLP2_CT=$m"
  HFTT_MLP2_PATCH=0 HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_ct$m.so python tools/bench_strip.py ffn
done
