#!/bin/bash
# Same-box A/B of the whole training step (VERDICT r05 rule: a structural kernel change is kept only on >= 3 interleaved runs each way).
#   bash tools/ab_step.sh <tagB> [rounds] [extra bench args]     A = the product library, B = nylon-amt_amd/lib/libhftt_hip_<tagB>.so
# (build B with HFTT_BUILD_TAG=<tagB> HFTT_BUILD_EXTRA_FLAGS="..." python nylon-amt_amd/build.py).  Prints clips/s and the per-step median of each run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; N=${2:-3}; shift; shift
B="python3 $R/bench.py --steps 30 --warmup 10 --no-extras --no-pmc --no-cpu-baseline --no-profile $*"
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = B ]; then export HFTT_LIB_PATH=$R/nylon-amt_amd/lib/libhftt_hip_$TAG.so; else unset HFTT_LIB_PATH; fi
    timeout -k 10 300 $B 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $i: %.1f clips/s  step ms min/median/max %.2f %.2f %.2f' % (j['value'], j['step_ms_min'], j['step_ms_median'], j['step_ms_max']))" || exit 1
  done
done
