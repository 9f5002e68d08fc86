#!/usr/bin/env python3
"""Dev tool (GPU box): gradient differences between HFTT_BF16_GRAD settings at a d=256 mini config, per tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import torch
import bench
from hftt_hip.trainer import TrainStep
dev = torch.device('cuda:0')
cfg = bench.BenchCfg(4, 16, 48, 4, 5, 256, 512, 2, 2, 4, 4, 12, 16)
B = 2
x, ld = bench.synthetic_batch(cfg, B, 21, dev)
def run(flag, p):
    os.environ['HFTT_BF16_GRAD'] = flag
    model = bench.build_model(cfg, 7, p, dev)
    model.hftt_precision = 'bf16'
    model.train()
    ts = TrainStep(model, lr=1e-3)
    ts.forward_backward(x, *ld)
    torch.cuda.synchronize()
    e = ts.engine
    return {n: e.flat_grads[o:o + k].clone() for (n, _, o, k) in e._bound}, e._ws[B]['bf16_grad'], e._ws[B]['seed']
def cmp(a, b, tag):
    rows = []
    for n, g in a.items():
        sc = g.abs().max().item()
        if sc < 1e-7: continue
        rows.append(((b[n] - g).abs().max().item() / sc, n))
    rows.sort(reverse=True)
    print(tag, 'worst:', ['%.3g %s' % r for r in rows[:6]])
for p in (0.0, 0.1):
    a, fa, sa = run('0', p); a2, _, sa2 = run('0', p); b, fb, sb = run('1', p)
    print('dropout', p, 'engaged', fa, fb, 'seeds', sa, sa2, sb)
    cmp(a, a2, '  0 vs 0 ')
    cmp(a, b, '  0 vs 1 ')
a, _, _ = run('0', 0.1); b, _, _ = run('1', 0.1)
for n in a:
    if n.endswith('fc_k.bias'):
        q = n.replace('fc_k.bias', 'fc_q.bias')
        print('%-62s |g| default %.3g  bf16-stream %.3g   (fc_q.bias scale %.3g)' % (n, a[n].abs().max().item(), b[n].abs().max().item(), a[q].abs().max().item()))
