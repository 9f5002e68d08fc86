#!/bin/bash
# round 5, GPU call 14: one-instruction ReLU -- strip tests (bit identity between the forms), x3 / bf16 FFN times, paper steps in both modes
tools/gpu_steps.sh \
 "strip_tests|500|python -m pytest tests/test_strip_gpu.py tests/test_x3_gpu.py -x -q" \
 "bf16_ffn|120|python tools/bench_strip.py ffn" \
 "x3_ffn|120|FFN_ONLY=1 python tools/bench_x3.py" \
 "bench_x3|200|python bench.py --no-cpu-baseline --no-extras --no-pmc --no-profile > gpurun_out/r05_x3_relu.json; tail -c 120 gpurun_out/r05_x3_relu.json"
