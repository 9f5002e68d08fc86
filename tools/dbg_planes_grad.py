"""x3 strip plans with q / k / v as f16-pair planes against the same plans on fp32 q / k / v (HFTT_X3_PLANES=0): outputs and every gradient of one
training step at the convergence test's WIDE configuration (and at short paper-like axes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import torch
import util
from util import O
from hftt_hip.trainer import TrainStep
dev = torch.device('cuda:0')
WIDE = O.HfttConfig(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, hid_dim=256, pf_dim=512, enc_layer=1, dec_layer=1,
                    enc_head=4, dec_head=4, n_note=8, n_velocity=16)
for name, cfg, B in (('wide', WIDE, 4),):
    x = O.synth_spec(B, cfg, salt=1)
    labels = O.synth_labels(B, cfg, salt=2)
    res = {}
    for planes in ('0', '1'):
        os.environ['HFTT_X3_PLANES'] = planes
        model = util.build_model(cfg, 2025, dropout=0.0).to(dev)
        model.hftt_precision = 'x3'
        model.train()
        ts = TrainStep(model)
        loss = ts.forward_backward(x.to(dev), *[t.to(dev).contiguous() for t in labels])
        torch.cuda.synchronize()
        eng = ts.engine
        res[planes] = ([o.clone() for o in eng._ws[B]['outs']], eng.flat_grads.clone(), float(loss[0]), eng)
    a, b = res['0'], res['1']
    print(name, 'loss', a[2], b[2])
    for i, (u, v) in enumerate(zip(a[0], b[0])):
        print('  out', i, float((u - v).abs().max()))
    eng = a[3]
    worst = []
    for (pname, _, o, n) in eng._bound:
        ga, gb = a[1][o:o + n], b[1][o:o + n]
        sc = float(ga.abs().max())
        if sc < 1e-12:
            continue
        worst.append((float((ga - gb).abs().max()) / sc, pname))
    worst.sort(reverse=True)
    for e, n_ in worst[:12]:
        print('  %.3e  %s' % (e, n_))
