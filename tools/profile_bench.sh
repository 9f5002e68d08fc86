#!/bin/bash
# GPU box: the default bench line + rocprofv3 kernel stats + PMC passes around bench.py; outputs under gpurun_out/<tag>_*.
#   pass 1  --kernel-trace --stats                      per-kernel durations (same command as the bench's timed leg, no event instrumentation)
#   pass 2  --pmc FETCH_SIZE      pass 3  --pmc WRITE_SIZE       HBM traffic (MI355X_MICROARCH.md: separate passes, FETCH_SIZE x2 on gfx950)
#   pass 4  --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE      pass 5  --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES (utilisation)
# Usage (inside gpurun): [BENCH_ARGS="--precision bf16"] bash tools/profile_bench.sh <tag> [quick]   then locally: python tools/save_profiles.py <tag> ["mode text"]
# BENCH_ARGS is appended to every bench.py / bench_inference.py command (precision mode, --config tiny).
# Each step runs only if the previous one ended normally (a killed step stops the script).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
B="python3 $R/bench.py $BENCH_ARGS"
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 "$@" > $O/${TAG}_$name.log 2>&1; rc=$?; if [ $rc -ne 0 ]; then echo "$name rc=$rc"; tail -n 5 $O/${TAG}_$name.log; fi; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then exit $rc; fi; }
if [ "$2" != "quick" ]; then
  echo "== bench"; timeout -k 10 420 $B > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; rc=$?; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then exit $rc; fi
fi
S="--no-cpu-baseline --no-profile --no-extras"
run stats rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o bench --output-format csv -- $B --steps 10 --warmup 3 $S
run fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_fetch -o b --output-format csv -- $B --steps 2 --warmup 1 $S
run write rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_write -o b --output-format csv -- $B --steps 2 --warmup 1 $S
run busy rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${TAG}_busy -o b --output-format csv -- $B --steps 2 --warmup 1 $S
run busy2 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $O/${TAG}_busy2 -o b --output-format csv -- $B --steps 2 --warmup 1 $S
# the inference plan (eval forward only): its own stats + traffic + busy passes
I="python3 $R/tools/bench_inference.py $BENCH_ARGS"
run inf_stats rocprofv3 --kernel-trace --stats -d $O/${TAG}_inf_stats -o bench --output-format csv -- $I --steps 10
run inf_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_inf_fetch -o b --output-format csv -- $I --steps 2
run inf_write rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_inf_write -o b --output-format csv -- $I --steps 2
run inf_busy rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${TAG}_inf_busy -o b --output-format csv -- $I --steps 2
if [ -f $O/${TAG}_bench.json ]; then tail -c 400 $O/${TAG}_bench.json; fi
exit 0
