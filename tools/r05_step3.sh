#!/bin/bash
# round 5, GPU call 3: position-table scale 300 on both model sizes; the rewritten convergence test (calibration run)
tools/gpu_steps.sh \
 "sweep_tiny_pos300|300|python tools/sweep_recipe.py --config tiny --steps 10000 --recipes pos300:3e-4:0:1:0:300 pos300w1e-3:1e-3:500:1:0:300 > gpurun_out/r05_sweep_tiny_pos300.json" \
 "sweep_paper_pos300|560|python tools/sweep_recipe.py --config paper --steps 6000 --recipes pos300:1e-4:0:1:0:300 pos300w3e-4:3e-4:500:1:0:300 > gpurun_out/r05_sweep_paper_pos300.json" \
 "convergence|500|python -m pytest tests/test_convergence_gpu.py -x -q -s"
