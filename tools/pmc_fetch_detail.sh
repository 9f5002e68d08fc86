#!/bin/bash
# GPU box: what the L2 really fetched, by request size.  FETCH_SIZE counts every L2 -> fabric read request at 64 bytes (gfx950: a wide streaming
# read goes out as 128-byte requests, hence the "x 2" of MI355X_MICROARCH.md -- which OVER-counts kernels whose reads leave as 64-byte or 32-byte
# requests).  The raw counters split the requests by size; bytes = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B.  Second pass: L2 hits / misses.
#   bash tools/pmc_fetch_detail.sh <tag>      then locally: python tools/pmc_fetch_detail.py <tag>
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
B="python3 $R/bench.py $BENCH_ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/${TAG}_rdreq -o b --output-format csv -- $B > $O/${TAG}_rdreq.log 2>&1 || { tail -5 $O/${TAG}_rdreq.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/${TAG}_l2 -o b --output-format csv -- $B > $O/${TAG}_l2.log 2>&1 || { tail -5 $O/${TAG}_l2.log; exit 1; }
echo done
