#!/usr/bin/env python3
"""dev: whole training step with the pipelined strip kernels (HFTT_STRIP_V2=1) vs the one-block form (=0): must be bit-identical."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
from util import O
from hftt_hip.trainer import TrainStep
dev = torch.device('cuda:0')
cfg = O.HfttConfig(n_margin=4, n_frame=16, n_bin=48, cnn_channel=4, cnn_kernel=5, hid_dim=256, pf_dim=512,
                   enc_layer=2, dec_layer=2, enc_head=4, dec_head=4, n_note=12, n_velocity=16)
B = 2
x = (O.synth_spec(B, cfg, salt=21) * 0.5).to(dev)
ld = tuple(t.to(dev).contiguous() for t in O.synth_labels(B, cfg, salt=22))
for drop in (0.0, 0.1):
    res = {}
    for v2 in ('0', '1'):
        os.environ['HFTT_STRIP_V2'] = v2
        model = util.build_model(cfg, 7, dropout=drop).to(dev)
        model.hftt_precision = 'bf16'
        model.train()
        ts = TrainStep(model, lr=1e-3)
        loss = ts.forward_backward(x, *ld)
        torch.cuda.synchronize()
        eng = ts.engine
        res[v2] = ([t.clone() for t in eng._ws[B]['outs']], {n: eng.flat_grads[o:o + k].clone() for (n, _, o, k) in eng._bound},
                   {k: v.clone() for k, v in eng._ws[B]['bufs'].items() if k.startswith('g.') or k.endswith('.x2') or k.endswith('.h')})
    print('dropout', drop, 'outputs equal:', all(torch.equal(a, b) for a, b in zip(res['0'][0], res['1'][0])))
    bad = [(n, (res['0'][1][n] - res['1'][1][n]).abs().max().item() / (res['0'][1][n].abs().max().item() + 1e-30)) for n in res['0'][1]
           if not torch.equal(res['0'][1][n], res['1'][1][n])]
    print('  gradient tensors that differ: %d of %d' % (len(bad), len(res['0'][1])), bad[:6])
    badb = [k for k in res['0'][2] if not torch.equal(res['0'][2][k], res['1'][2][k])]
    print('  workspace buffers that differ:', badb[:20])
