#!/bin/bash
# GPU box: the <= 128-key attention backward forms at two workgroups per CU (register bound), and the key-halves proxy again on them
set -e
cd "$(dirname "$0")/.."
python -m pytest tests/test_x3_gpu.py tests/test_attention_gpu.py -q -m gpu -k "attention or attn" 2>&1 | tail -2
NSEQ=704 LQ=128 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=1024 LQ=88 LK=88 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=1024 LQ=256 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=2048 LQ=256 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=1024 LQ=88 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=2048 LQ=88 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
