#!/usr/bin/env python3
"""dev: poison every workspace buffer with NaN, rerun a training step, report NaNs in outputs / gradients (reads of never-written memory)."""
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
    model = util.build_model(cfg, 7, dropout=drop).to(dev)
    model.hftt_precision = 'bf16'
    model.train()
    ts = TrainStep(model, lr=1e-3)
    ts.forward_backward(x, *ld)
    torch.cuda.synchronize()
    eng = ts.engine
    ref = eng.flat_grads.clone()
    ws = eng._ws[B]
    for k, t in ws['bufs'].items():
        if k == 'spec' or not t.is_floating_point():
            continue
        t.fill_(float('nan'))
    eng.flat_grads.fill_(float('nan'))
    eng.step_counter -= 1                      # same dropout seed as the first run
    ts.forward_backward(x, *ld)
    torch.cuda.synchronize()
    nan_g = [n for (n, _, o, k) in eng._bound if not torch.isfinite(eng.flat_grads[o:o + k]).all()]
    diff_g = [n for (n, _, o, k) in eng._bound if torch.isfinite(eng.flat_grads[o:o + k]).all() and not torch.equal(eng.flat_grads[o:o + k], ref[o:o + k])]
    print('dropout', drop, 'strip', eng.strip, ': gradient tensors with NaN:', len(nan_g), nan_g[:8])
    print('   finite but different from the first run:', len(diff_g), diff_g[:8])
    print('   outputs finite:', all(torch.isfinite(t).all().item() for t in ws['outs']))
