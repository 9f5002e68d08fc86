#!/bin/bash
# round 5, GPU call 9: whole-line LOADS of the bf16 strip kernels (tests, per-form A/B), bf16 step
tools/gpu_steps.sh \
 "strip_tests|400|python -m pytest tests/test_strip_gpu.py -x -q" \
 "lines_all|120|HFTT_MLP2_PATCH=1 python tools/bench_strip.py strip" \
 "lines_mlp0|120|HFTT_MLP2_PATCH=0 python tools/bench_strip.py ffn" \
 "bench_bf16|200|python bench.py --precision bf16 --no-cpu-baseline --no-extras --no-pmc --no-profile > gpurun_out/r05_bf16_lines.json; tail -c 200 gpurun_out/r05_bf16_lines.json" \
 "model_tests|500|python -m pytest tests/test_model_gpu.py tests/test_paper_bf16_gpu.py -x -q"
