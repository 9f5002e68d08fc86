#!/bin/bash
# round 5, GPU call 7: whole-line stores of the bf16 fused FFN (tests + A/B), the two register plans element for element, the whole GPU suite
tools/gpu_steps.sh \
 "ffn_tests|300|python -m pytest tests/test_strip_gpu.py -x -q -k 'ffn'" \
 "ffn_patch0|120|HFTT_MLP2_PATCH=0 python tools/bench_strip.py strip" \
 "ffn_patch1|120|HFTT_MLP2_PATCH=1 python tools/bench_strip.py strip" \
 "dbg_wpc|120|python tools/dbg_mlp2_wpc.py" \
 "gpu_suite|800|python -m pytest tests -q -m gpu"
