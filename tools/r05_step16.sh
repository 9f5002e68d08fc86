#!/bin/bash
# GPU box: attention tests on the interleaved-dQ backward, then its timing on the encoder / cross shapes (standalone)
set -e
cd "$(dirname "$0")/.."
python -m pytest tests/test_x3_gpu.py -q -m gpu -k "attention" 2>&1 | tail -3
NSEQ=1024 LQ=256 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=1024 LQ=88 LK=256 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
NSEQ=704 LQ=128 LK=128 python tools/bench_attn_bwd_shapes.py 2>&1 | tail -1
