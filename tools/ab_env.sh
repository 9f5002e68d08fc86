#!/bin/bash
# Same-box A/B of the whole training step between two settings of ONE environment variable (same library):
#   bash tools/ab_env.sh VAR valueA valueB [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; VA=$2; VB=$3; N=${4:-3}; shift; shift; shift; shift
B="python3 $R/bench.py --steps 30 --warmup 10 --no-extras --no-pmc --no-cpu-baseline --no-profile $*"
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = B ]; then export $VAR=$VB; else export $VAR=$VA; fi
    timeout -k 10 300 $B 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v ($VAR=' + '$( [ $v = B ] && echo $VB || echo $VA )' + ') round $i: %.1f clips/s  step ms min/median/max %.2f %.2f %.2f' % (j['value'], j['step_ms_min'], j['step_ms_median'], j['step_ms_max']))" || exit 1
  done
done
