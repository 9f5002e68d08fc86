#!/bin/bash
# A/B on one box: the product library against a build of x3_strip.hip with -DHFTT_X3_G8=1 (gradient strips contribute their hi half only).
set -e
cd "$(dirname "$0")/../nylon-amt_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DHFTT_X3_G8=1 -x hip -c csrc/x3_strip.hip -o build/x3_strip_g8.o
OBJS=$(ls build/*.o | grep -v "x3_strip\|\.x\.o\|strip_gemm[345]")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhftt_g8.so $OBJS build/x3_strip_g8.o
cd ..
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | cut -c 60-140
  HFTT_TN_DY_HI=1 HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_g8.so python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | cut -c 60-140
done
HFTT_TN_DY_HI=1 HFTT_LIB_PATH=$PWD/nylon-amt_amd/lib/libhftt_g8.so python -m pytest tests/test_paper_bf16_gpu.py -q -m gpu -k "x3_mode" -s 2>&1 | grep -E "x3 mode vs|passed|failed|assert" | cut -c 1-1800
