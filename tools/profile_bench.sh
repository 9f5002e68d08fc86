#!/bin/bash
# GPU box: rocprofv3 kernel stats + two PMC passes (FETCH_SIZE, WRITE_SIZE) around bench.py; outputs under gpurun_out/<tag>_*.
# Usage (inside gpurun): bash tools/profile_bench.sh <tag>      then locally: python tools/save_profiles.py <tag> gpurun_out/<tag>_stats
#                        gpurun_out/<tag>_bench.json gpurun_out/<tag>_fetch gpurun_out/<tag>_write
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o bench --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o b --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o b --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/${TAG}_write.log 2>&1
tail -c 600 $R/gpurun_out/${TAG}_bench.json
