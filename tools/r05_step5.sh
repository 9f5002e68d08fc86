#!/bin/bash
# round 5, GPU call 5: new tests (trained paper-size weights vs the reference, valid(metrics=True) replay, convergence), FFN ablation + counters
tools/gpu_steps.sh \
 "t_paper_trained|300|python -m pytest tests/test_paper_bf16_gpu.py -x -q -s -k trained" \
 "t_replay|200|python -m pytest tests/test_train_replay_gpu.py -x -q -k 'metrics or replay'" \
 "t_convergence|600|python -m pytest tests/test_convergence_gpu.py -q -s" \
 "ablate_mlp2|400|tools/ablate_mlp2.sh" \
 "pmc_ffn|300|tools/pmc_ffn.sh"
