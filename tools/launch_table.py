#!/usr/bin/env python3
"""Dev tool (GPU box): per-launch table of one paper-size training step (B=8), grouped by (kernel, shape, bytes).
Usage: python tools/launch_table.py [--precision bf16|parity] [--steps N] [--raw]"""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import torch
import bench                      # workload tables / model builder / synthetic clips of the measured leg (no oracle)
from hftt_hip.trainer import TrainStep
from hftt_hip.profiler import LaunchProfiler

ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='bf16')
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--raw', action='store_true', help='print every launch of one step in plan order')
args = ap.parse_args()

dev = torch.device('cuda:0')
cfg, B = bench.CONFIGS['paper'], args.batch
model = bench.build_model(cfg, 1234, 0.1, dev)
model.hftt_precision = args.precision
model.train()
ts = TrainStep(model, lr=1e-4)
x, lab = bench.synthetic_batch(cfg, B, 1, dev)
for _ in range(3):
    ts(x, *lab)
torch.cuda.synchronize()
prof = LaunchProfiler()
ts.engine.profiler = prof
for _ in range(args.steps):
    ts.forward_backward(x, *lab)
torch.cuda.synchronize()
recs = [(n, m, e0.elapsed_time(e1) * 1e3) for n, m, e0, e1 in prof.records]
ts.engine.profiler = None
per_step = len(recs) // args.steps
if args.raw:
    for i, (n, m, us) in enumerate(recs[-per_step:]):
        print('%4d %-42s %-22s %8.1f us %8.1f MB' % (i, m['kernel'] if m else n, m.get('shape', '') if m else '', us, (m['bytes'] / 1e6) if m else 0))
groups = {}
for n, m, us in recs:
    key = (m['kernel'], str(m.get('shape', '')), round(m['bytes'] / 1e6)) if m else (n, '', 0)
    g = groups.setdefault(key, [0, 0.0, m])
    g[0] += 1
    g[1] += us
tot = sum(g[1] for g in groups.values()) / args.steps
print('%-44s %-24s %6s %9s %8s %8s %8s %8s' % ('kernel', 'shape', 'n/step', 'avg us', 'MB', 'GB/s', 'TF/s', 'ms/step'))
for key, (cnt, us, m) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    avg = us / cnt
    print('%-44s %-24s %6.1f %9.1f %8d %8.0f %8.1f %8.3f' % (key[0][:44], key[1], cnt / args.steps, avg, key[2], key[2] * 1e3 / avg if key[2] else 0,
                                                             (m['flops'] / avg / 1e6) if m else 0, us / args.steps / 1e3))
print('sum of launches: %.2f ms/step over %d launches' % (tot / 1e3, per_step))
