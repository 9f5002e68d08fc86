#!/bin/bash
# round 5, GPU call 2: the two-workgroups-per-CU fused FFN (tests + A/B), position-table scale on the small and the paper-size model
tools/gpu_steps.sh \
 "ffn_tests|400|python -m pytest tests/test_strip_gpu.py -x -q -k 'ffn'" \
 "ffn_wpc1|200|HFTT_MLP2_WPC=1 HFTT_MLP2_WPC_BWD=1 python tools/bench_strip.py strip" \
 "ffn_wpc2|200|HFTT_MLP2_WPC=2 HFTT_MLP2_WPC_BWD=2 python tools/bench_strip.py strip" \
 "sweep_tiny_pos|300|python tools/sweep_recipe.py --config tiny --steps 10000 --recipes pos30:3e-4:0:1:0:30 pos30w1e-3:1e-3:500:1:0:30 > gpurun_out/r05_sweep_tiny_pos.json" \
 "sweep_paper_pos|520|python tools/sweep_recipe.py --config paper --steps 4000 --recipes pos30:1e-4:0:1:0:30 pos30w3e-4:3e-4:500:1:0:30 pos100w3e-4:3e-4:500:1:0:100 > gpurun_out/r05_sweep_paper_pos.json"
