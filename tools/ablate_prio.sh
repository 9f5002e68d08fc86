#!/bin/bash
# GPU box: build libhftt_prio.so with -DHFTT_PRIO_SKEW=<n> on the two attention sources (a static priority for the first half of the waves of
# an 8-wave workgroup), time the attention launches and one bench line with and without it.
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
for f in x3_attn x3_attn_pl; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_PRIO_SKEW=${PRIO:-1} -x hip -c csrc/$f.hip -o build/${f}_prio_ablate.o &
done
wait
OBJS=$(ls build/*.o | grep -v "/x3_attn\.o\|/x3_attn_pl\.o\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_prio.so $OBJS build/x3_attn_prio_ablate.o build/x3_attn_pl_prio_ablate.o
cd ..
echo "== default"; python tools/bench_x3_attn.py 2>/dev/null
echo "== priority skew"; HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_prio.so python tools/bench_x3_attn.py 2>/dev/null
