#!/bin/bash
# GPU box: (1) the key-halves proxy -- the 128-key backward (two 4-wave workgroups per CU) on twice the sequences against the 256-key kernel;
# (2) timing switches of the plane backward: 64 = dQ product off (dS still written), 16 = dS copy + dQ off, 1 = staging conversions off
set -e
cd "$(dirname "$0")/.."
echo "== key-halves proxy (product library)"
NSEQ=1024 LQ=256 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -2
NSEQ=2048 LQ=256 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -2
NSEQ=1024 LQ=88 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -2
NSEQ=2048 LQ=88 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -2
echo "== timing switches (ablation build)"
cd nylon-amt_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_X3_ATTN_ABLATE -x hip -c csrc/x3_attn_pl.hip -o build/x3_attn_pl_ablate.o
OBJS=$(ls build/*.o | grep -v "x3_attn_pl\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_xattn.so $OBJS build/x3_attn_pl_ablate.o
cd ..
for bits in ${ABLATE_BITS:-0 64 16 1 2 8 65}; do
  HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_xattn.so HFTT_X3_ATTN_DEBUG=$bits NSEQ=1024 LQ=256 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -2
done
