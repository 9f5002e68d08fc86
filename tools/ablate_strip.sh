#!/bin/bash
# Build libhftt_ablate.so (strip_gemm.hip with -DHFTT_STRIP_ABLATE, the other objects as built) and time the strip kernels with single
# mechanisms switched off (see the ABL comment in csrc/strip_gemm.hip).  Run on the GPU box AFTER nylon-amt_amd/build.py.
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_STRIP_ABLATE -x hip -c csrc/strip_gemm.hip -o build/strip_gemm_ablate.o
OBJS=$(ls build/*.o | grep -v strip_gemm)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_ablate.so $OBJS build/strip_gemm_ablate.o
cd ..
for bits in ${ABLATE_BITS:-0 1 2 3 4 8 6 7 15}; do
  echo "### HFTT_STRIP_ABLATE=$bits"
  HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_ablate.so HFTT_STRIP_ABLATE=$bits python tools/bench_strip.py strip
done
