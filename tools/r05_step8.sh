#!/bin/bash
# round 5, GPU call 8: whole-line stores of the bf16 linear strip kernel (tests + A/B), the bf16-mode step, the whole GPU suite
tools/gpu_steps.sh \
 "strip_tests|400|python -m pytest tests/test_strip_gpu.py -x -q" \
 "lin_patch0|120|HFTT_LINEAR2_PATCH=0 python tools/bench_strip.py strip" \
 "lin_patch1|120|HFTT_LINEAR2_PATCH=1 python tools/bench_strip.py strip" \
 "bench_bf16_p0|200|HFTT_LINEAR2_PATCH=0 HFTT_MLP2_PATCH=0 python bench.py --precision bf16 --no-cpu-baseline --no-extras --no-pmc --no-profile > gpurun_out/r05_bf16_p0.json; tail -c 300 gpurun_out/r05_bf16_p0.json" \
 "bench_bf16_p1|200|python bench.py --precision bf16 --no-cpu-baseline --no-extras --no-pmc --no-profile > gpurun_out/r05_bf16_p1.json; tail -c 300 gpurun_out/r05_bf16_p1.json" \
 "gpu_suite|800|python -m pytest tests -q -m gpu"
