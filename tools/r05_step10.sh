#!/bin/bash
# round 5, GPU call 10: final build -- whole GPU suite, strip bench, bf16 step
tools/gpu_steps.sh \
 "gpu_suite|800|python -m pytest tests -q -m gpu" \
 "strip|120|python tools/bench_strip.py strip" \
 "bench_bf16|200|python bench.py --precision bf16 --no-cpu-baseline --no-extras --no-pmc --no-profile > gpurun_out/r05_bf16_final.json; tail -c 200 gpurun_out/r05_bf16_final.json"
