#!/bin/bash
# GPU box: the x3 weight-gradient GEMMs with and without the operand split in their loader (libhftt_tn_nosplit.so: gemm_tn.hip with
# -DHFTT_TN_NOSPLIT, the other objects as built by nylon-amt_amd/build.py; results of that build are garbage, times are the point).
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_TN_NOSPLIT -x hip -c csrc/gemm_tn.hip -o build/gemm_tn_ablate.o
OBJS=$(ls build/*.o | grep -v "gemm_tn\|\.x\.o\|\.g\.o\|strip_gemm[345]\|_g8\|_ablate")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_tn_nosplit.so $OBJS build/gemm_tn_ablate.o
cd ..
echo "### product loader (splits both fp32 operands)"
python tools/bench_tn_x3.py
echo "### loader without the split"
HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_tn_nosplit.so python tools/bench_tn_x3.py
