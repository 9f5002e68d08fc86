#!/usr/bin/env python3
"""Inference plan only (model.eval() forward, default: paper-size model, batch 8, x3 mode): the command the inference-side rocprofv3 passes
of tools/profile_bench.sh wrap.  Prints clips/s."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'nylon-amt_amd')]
import torch   # noqa: E402
import bench   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--precision', default='x3', choices=['x3', 'bf16', 'parity'])
ap.add_argument('--config', default='paper', choices=['paper', 'tiny'])
args = ap.parse_args()
dev = torch.device('cuda', 0)
cfg = bench.CONFIGS[args.config]
model = bench.build_model(cfg, 1234, 0.1, dev)
model.hftt_precision = args.precision
model.eval()
model.hftt_freeze_weights(True)
x, _ = bench.synthetic_batch(cfg, args.batch, 1234, dev)
with torch.no_grad():
    for _ in range(2):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model(x)
    torch.cuda.synchronize()
print('inference clips/s %.1f' % (args.batch * args.steps / (time.perf_counter() - t0)))
