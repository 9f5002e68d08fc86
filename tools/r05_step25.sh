#!/bin/bash
# GPU box: the attention backward shapes again in the steady state (WARM=300 launches before the timed ones)
cd "$(dirname "$0")/.."
export WARM=300
NSEQ=1024 LQ=256 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=2048 LQ=256 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=1024 LQ=88 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=2048 LQ=88 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=704 LQ=128 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=1024 LQ=88 LK=88 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
