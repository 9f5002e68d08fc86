#!/bin/bash
# round 5, GPU call 11: the bf16 small-width family in the engine -- model tests, tiny bench in both modes
tools/gpu_steps.sh \
 "model_tests|600|python -m pytest tests/test_model_gpu.py tests/test_paper_bf16_gpu.py tests/test_convergence_gpu.py -x -q" \
 "tiny_bf16|200|python bench.py --config tiny --precision bf16 --no-cpu-baseline --no-extras --no-pmc > gpurun_out/r05_tiny_bf16_new.json; tail -c 150 gpurun_out/r05_tiny_bf16_new.json" \
 "tiny_x3|200|python bench.py --config tiny --no-cpu-baseline --no-extras --no-pmc > gpurun_out/r05_tiny_x3_new.json; tail -c 150 gpurun_out/r05_tiny_x3_new.json"
