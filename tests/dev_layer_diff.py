#!/usr/bin/env python3
"""Dev tool (GPU box): compare every engine activation buffer against the CPU oracle's intermediates, layer by layer,
and every gradient, to localise numerical error.  Usage: python tests/dev_layer_diff.py [tiny|paper|mini] [parity|bf16]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
import util
from util import O

which = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
prec = sys.argv[2] if len(sys.argv) > 2 else 'parity'
cfg = {'tiny': O.TINY, 'paper': O.PAPER, 'mini': util.MINI}[which]
B = 1 if which == 'paper' else 2
dev = torch.device('cuda:0')
model = util.build_model(cfg, 4321); util.perturb(model, 4322)
sd = util.sd_cpu(model)
x = O.synth_spec(B, cfg, salt=4321); labels = O.synth_labels(B, cfg, salt=4328)

# ---- oracle with intermediates (double precision for a clean reference) ----
sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
inter = {}
def enc_layer(pre, xx, H, tag):
    a, _ = O.mha(sd64, pre + 'self_attention.', xx, xx, xx, H)
    x1 = O._ln(sd64, pre, xx + a); inter[tag + '.x1'] = x1
    f = O.ffn(sd64, pre + 'positionwise_feedforward.', x1)
    x2 = O._ln(sd64, pre, x1 + f); inter[tag + '.x2'] = x2
    return x2
e = 'encoder_spec2midi.'
T, Fq, d, N, V = cfg.n_frame, cfg.n_bin, cfg.hid_dim, cfg.n_note, cfg.n_velocity
win = x.double().unfold(2, cfg.n_proc, 1).permute(0, 2, 1, 3).contiguous().reshape(B * T, 1, Fq, cfg.n_proc)
cnn = F.conv2d(win, sd64[e + 'conv.weight'], sd64[e + 'conv.bias']).permute(0, 2, 1, 3).contiguous().reshape(B * T, Fq, cfg.cnn_dim)
tok = F.linear(cnn, sd64[e + 'tok_embedding_freq.weight'], sd64[e + 'tok_embedding_freq.bias'])
xx = tok * math.sqrt(d) + sd64[e + 'pos_embedding_freq.weight'][:Fq].unsqueeze(0); inter['x0'] = xx
for i in range(cfg.enc_layer):
    xx = enc_layer(f'{e}layers_freq.{i}.', xx, cfg.enc_head, f'enc{i}')
out = O.decoder_forward(sd64, xx.reshape(B, T, Fq, d), cfg)
loss = O.spec2midi_loss(out, *labels)
loss.backward()

from hftt_hip.trainer import TrainStep
model = model.to(dev); model.hftt_precision = prec; model.train()
ts = TrainStep(model)
l = ts.forward_backward(x.to(dev), *[t.to(dev).contiguous() for t in labels])
ws = ts.engine._ws[B]; b = ws['bufs']
def rep(name, mine, ref):
    ref = ref.detach().double().cpu().reshape(-1); mine = mine.detach().double().cpu().reshape(-1)
    err = (mine - ref).abs().max().item(); sc = ref.abs().max().item()
    print(f'{name:60s} max|ref|={sc:10.4g} max_err={err:10.3g} rel={err / max(sc, 1e-30):9.3g}')
print('loss', l[0].item(), loss.item())
rep('x0', b['x0'], inter['x0'])
for i in range(cfg.enc_layer):
    rep(f'enc{i}.x1', b[f'enc{i}.x1'], inter[f'enc{i}.x1']); rep(f'enc{i}.x2', b[f'enc{i}.x2'], inter[f'enc{i}.x2'])
for n, t, r in zip(util.OUT_NAMES, ws['outs'], out):
    rep('out.' + n, t, r)
print('---- gradients (sorted by rel err) ----')
rows = []
for (name, _, o, n) in ts.engine._bound:
    g = ts.engine.flat_grads[o:o + n].double().cpu(); r = sd64[name].grad.reshape(-1)
    err = (g - r).abs().max().item(); sc = r.abs().max().item()
    rows.append((err / max(sc, 1e-30), name, sc, err))
for r_ in sorted(rows, reverse=True)[:25]:
    print(f'{r_[1]:70s} max|ref|={r_[2]:10.4g} err={r_[3]:10.3g} rel={r_[0]:9.3g}')
