#!/bin/bash
# round 5, GPU call 4: the paper-size model trained by this path (position tables x 300, warm-up to 3e-4, cosine) + same-checkpoint A/B of the
# precision modes at two points of that run
S=/tmp/hftt_state_%d.pt
tools/gpu_steps.sh \
 "train_paper|470|python tools/train_config5.py --config paper --steps 12000 --lr 3e-4 --warmup 500 --final-frac 0.1 --pos-scale 300 --save-state $S --save-state-at 3000,10000 --out gpurun_out/config5_paper.pkl > gpurun_out/r05_config5_paper_trained.json" \
 "ab3k_x3|120|python tools/ab_modes.py --state /tmp/hftt_state_3000.pt --mode x3 --steps 1500 --out /tmp/ab3_x3.pt" \
 "ab3k_bf16|120|python tools/ab_modes.py --state /tmp/hftt_state_3000.pt --mode bf16 --steps 1500 --out /tmp/ab3_bf16.pt" \
 "ab3k_x3h|120|HFTT_X3_FP32_HIDDEN=1 python tools/ab_modes.py --state /tmp/hftt_state_3000.pt --mode x3 --steps 1500 --out /tmp/ab3_x3h.pt" \
 "ab3k_parity|260|python tools/ab_modes.py --state /tmp/hftt_state_3000.pt --mode parity --steps 1500 --out /tmp/ab3_parity.pt" \
 "ab3k|60|python tools/ab_modes.py --compare parity=/tmp/ab3_parity.pt x3=/tmp/ab3_x3.pt bf16=/tmp/ab3_bf16.pt x3_fp32_hidden=/tmp/ab3_x3h.pt > gpurun_out/r05_ab_step3000.json" \
 "ab10k_x3|100|python tools/ab_modes.py --state /tmp/hftt_state_10000.pt --mode x3 --steps 750 --out /tmp/ab10_x3.pt" \
 "ab10k_bf16|100|python tools/ab_modes.py --state /tmp/hftt_state_10000.pt --mode bf16 --steps 750 --out /tmp/ab10_bf16.pt" \
 "ab10k_parity|160|python tools/ab_modes.py --state /tmp/hftt_state_10000.pt --mode parity --steps 750 --out /tmp/ab10_parity.pt" \
 "ab10k|60|python tools/ab_modes.py --compare parity=/tmp/ab10_parity.pt x3=/tmp/ab10_x3.pt bf16=/tmp/ab10_bf16.pt > gpurun_out/r05_ab_step10000.json"
