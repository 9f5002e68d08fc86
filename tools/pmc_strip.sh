#!/bin/bash
# GPU box: PMC counters of the strip kernels at paper shapes (tools/bench_strip.py strip).  Three separate passes, as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass); the program sits directly behind `--`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_strip
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 $R/tools/bench_strip.py strip > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 $R/tools/bench_strip.py strip > $OUT/write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq -o p --output-format csv -- python3 $R/tools/bench_strip.py strip > $OUT/sq.log 2>&1 || exit 1
python3 $R/tools/pmc_strip_report.py $OUT
