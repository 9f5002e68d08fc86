#!/usr/bin/env python3
"""dev: gradient cosine matrix between builds (parity / round-1 bf16 / strip bf16) run in different orders in one process."""
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
skip = ('conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq', 'encoder_spec2midi.layers_freq.0.self_attention')


def run(build, drop):
    os.environ['HFTT_STRIP'] = '1' if build == 'strip' else '0'
    model = util.build_model(cfg, 7, dropout=drop).to(dev)
    model.hftt_precision = 'parity' if build == 'parity' else 'bf16'
    model.train()
    ts = TrainStep(model, lr=1e-3)
    ts.forward_backward(x, *ld)
    torch.cuda.synchronize()
    eng = ts.engine
    return {n: eng.flat_grads[o:o + k].clone().double() for (n, _, o, k) in eng._bound if not n.endswith('fc_k.bias') and not any(t in n for t in skip)}


def cos(a, b):
    va = torch.cat([a[k] for k in a]); vb = torch.cat([b[k] for k in a])
    return float(va @ vb / (va.norm() * vb.norm()))


for drop in (0.0, 0.1):
    s1 = run('strip', drop); p = run('parity', drop); r = run('round1', drop); s2 = run('strip', drop)
    print('dropout %.1f: strip(first) vs strip(last) %.5f | strip vs round1 %.5f | strip vs parity %.5f | round1 vs parity %.5f' %
          (drop, cos(s1, s2), cos(s1, r), cos(s1, p), cos(r, p)))
    worst = sorted((float(s1[k] @ p[k] / (s1[k].norm() * p[k].norm() + 1e-300)), float(s1[k].norm()), float(p[k].norm()), k) for k in s1)[:6]
    for w in worst:
        print('    cos %.4f  |strip| %.3e  |parity| %.3e  %s' % w)
