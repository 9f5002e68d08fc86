#!/bin/bash
# usage (GPU box): bash tools/prof_quick.sh <tag> [bench args...]: rocprofv3 kernel stats of a short bench run, top kernels printed
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o bench --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-extras "$@" > $R/gpurun_out/${TAG}_stats.log 2>&1
rc=$?
grep '"metric"' $R/gpurun_out/${TAG}_stats.log | cut -c1-260
f=$(find $R/gpurun_out/${TAG}_stats -name '*kernel_stats.csv' | sort | tail -n 1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print('%-90s calls %5s avg %10.1f us  %5s%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
exit $rc
