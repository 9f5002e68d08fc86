#!/bin/bash
# GPU box: the bf16-mode 88- and 128-key attention backward with / without the second launch bound (attn_bwd.hip built both ways)
set -e
cd "$(dirname "$0")/.."
for v in cap nocap; do
  if [ $v = nocap ]; then
    cd nylon-amt_amd
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_ATTN_BWD_NOCAP -x hip -c csrc/attn_bwd.hip -o build/attn_bwd_nocap.o
    OBJS=$(ls build/*.o | grep -v "/attn_bwd\.o\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate\|_stamps\|_nocap")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_nocap.so $OBJS build/attn_bwd_nocap.o
    cd ..
    export HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_nocap.so
  fi
  echo "== $v"
  NSEQ=704 LQ=128 LK=128 python tools/bench_attn_bwd_bf16.py 2>&1 | tail -1
  NSEQ=1024 LQ=88 LK=88 python tools/bench_attn_bwd_bf16.py 2>&1 | tail -1
done
