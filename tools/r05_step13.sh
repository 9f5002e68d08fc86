#!/bin/bash
# round 5, GPU call 13: tiny model in the bf16 mode on the plateau -- the old block-GEMM path (HFTT_STRIP=0) against the new strip family, same recipe
tools/gpu_steps.sh \
 "tiny_bf16_old|300|HFTT_STRIP=0 python tools/sweep_recipe.py --config tiny --precision bf16 --steps 20000 --log-every 1000 --recipes oldpath:3e-4:0:1:0:300 > gpurun_out/r05_tiny_bf16_oldpath.json" \
 "tiny_bf16_new|300|python tools/sweep_recipe.py --config tiny --precision bf16 --steps 30000 --log-every 1000 --recipes newpath:3e-4:0:1:0:300 > gpurun_out/r05_tiny_bf16_newpath.json"
