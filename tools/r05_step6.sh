#!/bin/bash
# round 5, GPU call 6: the whole GPU suite, the priority switch of the two-workgroup FFN, the default bench line
tools/gpu_steps.sh \
 "gpu_suite|700|python -m pytest tests -x -q -m gpu" \
 "ablate_prio|200|ABLATE_BITS='0 512' tools/ablate_mlp2.sh" \
 "bench_default|330|python bench.py > gpurun_out/r05a_bench_default.json 2> gpurun_out/r05a_bench_default.err; tail -c 1500 gpurun_out/r05a_bench_default.json"
