#!/bin/bash
# round 5, GPU call 12: does the bf16 throughput mode TRAIN the plucked-string corpus end to end?  tiny (the new small-width bf16 family) and paper size
tools/gpu_steps.sh \
 "train_tiny_bf16|300|python tools/train_config5.py --config tiny --precision bf16 --steps 15000 --lr 3e-4 --pos-scale 300 --out gpurun_out/config5_tiny_bf16.pkl > gpurun_out/r05_config5_tiny_bf16_trained.json" \
 "train_paper_bf16|500|python tools/train_config5.py --config paper --precision bf16 --steps 12000 --lr 3e-4 --warmup 500 --final-frac 0.1 --pos-scale 300 --out /tmp/config5_paper_bf16.pkl > gpurun_out/r05_config5_paper_bf16_trained.json"
