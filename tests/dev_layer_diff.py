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

# ---- oracle with intermediates: double precision as the clean reference, and the same graph in fp32 on the CPU as the yardstick ----
e = 'encoder_spec2midi.'
T, Fq, d, N, V = cfg.n_frame, cfg.n_bin, cfg.hid_dim, cfg.n_note, cfg.n_velocity
def oracle(dt):
    sdx = {k: v.to(dt).requires_grad_(True) for k, v in sd.items()}
    it = {}
    def enc_layer(pre, xx, H, tag):
        a, _ = O.mha(sdx, pre + 'self_attention.', xx, xx, xx, H)
        it[tag + '.attn'] = a
        x1 = O._ln(sdx, pre, xx + a); it[tag + '.x1'] = x1
        f = O.ffn(sdx, pre + 'positionwise_feedforward.', x1)
        x2 = O._ln(sdx, pre, x1 + f); it[tag + '.x2'] = x2
        return x2
    win = x.to(dt).unfold(2, cfg.n_proc, 1).permute(0, 2, 1, 3).contiguous().reshape(B * T, 1, Fq, cfg.n_proc)
    cnn = F.conv2d(win, sdx[e + 'conv.weight'], sdx[e + 'conv.bias']).permute(0, 2, 1, 3).contiguous().reshape(B * T, Fq, cfg.cnn_dim)
    tok = F.linear(cnn, sdx[e + 'tok_embedding_freq.weight'], sdx[e + 'tok_embedding_freq.bias'])
    xx = tok * math.sqrt(d) + sdx[e + 'pos_embedding_freq.weight'][:Fq].unsqueeze(0); it['x0'] = xx
    for i in range(cfg.enc_layer):
        xx = enc_layer(f'{e}layers_freq.{i}.', xx, cfg.enc_head, f'enc{i}')
    o = O.decoder_forward(sdx, xx.reshape(B, T, Fq, d), cfg)
    ls = O.spec2midi_loss(o, *labels)
    ls.backward()
    return sdx, it, o, ls
sd64, inter, out, loss = oracle(torch.float64)
sd32, inter32, out32, loss32 = oracle(torch.float32)

from hftt_hip.trainer import TrainStep
model = model.to(dev); model.hftt_precision = prec; model.train()
ts = TrainStep(model)
l = ts.forward_backward(x.to(dev), *[t.to(dev).contiguous() for t in labels])
ws = ts.engine._ws[B]; b = ws['bufs']
def rep(name, mine, ref, cpu32=None):
    ref = ref.detach().double().cpu().reshape(-1); mine = mine.detach().double().cpu().reshape(-1)
    err = (mine - ref).abs().max().item(); sc = ref.abs().max().item()
    extra = ''
    if cpu32 is not None:
        e32 = (cpu32.detach().double().cpu().reshape(-1) - ref).abs().max().item()
        extra = f' | cpu fp32 rel={e32 / max(sc, 1e-30):9.3g} ratio={err / max(e32, 1e-30):6.1f}'
    print(f'{name:60s} max|ref|={sc:10.4g} max_err={err:10.3g} rel={err / max(sc, 1e-30):9.3g}' + extra)
print('loss', l[0].item(), loss.item())
rep('x0', b['x0'], inter['x0'], inter32['x0'])
for i in range(cfg.enc_layer):
    rep(f'enc{i}.x1', b[f'enc{i}.x1'], inter[f'enc{i}.x1'], inter32[f'enc{i}.x1']); rep(f'enc{i}.x2', b[f'enc{i}.x2'], inter[f'enc{i}.x2'], inter32[f'enc{i}.x2'])
for n, t, r, r32 in zip(util.OUT_NAMES, ws['outs'], out, out32):
    rep('out.' + n, t, r, r32)
print('---- gradients (sorted by rel err) ----')
rows = []
for (name, _, o, n) in ts.engine._bound:
    g = ts.engine.flat_grads[o:o + n].double().cpu(); r = sd64[name].grad.reshape(-1)
    err = (g - r).abs().max().item(); sc = r.abs().max().item()
    e32 = (sd32[name].grad.reshape(-1).double() - r).abs().max().item()
    rows.append((err / max(sc, 1e-30), name, sc, err, e32 / max(sc, 1e-30)))
for r_ in sorted(rows, reverse=True)[:25]:
    print(f'{r_[1]:70s} max|ref|={r_[2]:10.4g} err={r_[3]:10.3g} rel={r_[0]:9.3g} | cpu fp32 rel={r_[4]:9.3g}')

# ---- where does the first layer's noise enter?  (parity mode: 'g.eq' still holds layer 0's dQ|dK|dV, 'x0' its input) ----
if prec == 'parity' and 'g.eq' in b:
    dq = b['g.eq'].double().cpu()[:, :d]; x0d = b['x0'].double().cpu().reshape(-1, d)
    name = e + 'layers_freq.0.self_attention.fc_q.weight'
    o_, n_ = [(o, n) for (nm, _, o, n) in ts.engine._bound if nm == name][0]
    g_dev = ts.engine.flat_grads[o_:o_ + n_].double().cpu().reshape(d, d)
    g_from_bufs = dq.T @ x0d
    r64 = sd64[name].grad
    sc = r64.abs().max().item()
    print('fc_q.weight: device TN vs fp64 product of the device buffers   rel=%.3g' % ((g_dev - g_from_bufs).abs().max().item() / sc))
    print('fc_q.weight: fp64 product of the device buffers vs fp64 oracle rel=%.3g' % ((g_from_bufs - r64).abs().max().item() / sc))
    g32 = (b['g.eq'].cpu()[:, :d].T @ b['x0'].cpu().reshape(-1, d)).double()
    print('fc_q.weight: CPU fp32 product of the device buffers vs their fp64 product rel=%.3g' % ((g32 - g_from_bufs).abs().max().item() / sc))

# ---- optional dump of layer 0's attention backward operands (first 8 sequences) for offline emulation: HFTT_DUMP_ATTN0=path.npz ----
if prec == 'parity' and os.environ.get('HFTT_DUMP_ATTN0') and 'enc0.qkv' in b:
    import numpy as np
    ns, L = 8, Fq
    np.savez_compressed(os.environ['HFTT_DUMP_ATTN0'],
                        qkv=b['enc0.qkv'].float().cpu().numpy().reshape(-1, L, 3 * d)[:ns], ctx=b['enc0.ctx'].float().cpu().numpy().reshape(-1, L, d)[:ns],
                        lse=b['enc0.lse'].float().cpu().numpy().reshape(-1)[:ns * cfg.enc_head * L * 2], dout=b['g.ex'].float().cpu().numpy().reshape(-1, L, d)[:ns],
                        dqkv=b['g.eq'].float().cpu().numpy().reshape(-1, L, 3 * d)[:ns], heads=cfg.enc_head)
    print('dumped', os.environ['HFTT_DUMP_ATTN0'])

# ---- is layer 0's Q | K | V projection as accurate as a CPU sgemm?  (same device input x0, fp64 product as the reference) ----
if prec == 'parity' and 'enc0.qkv' in b:
    pa = e + 'layers_freq.0.self_attention.'
    x0d = b['x0'].float().cpu().reshape(-1, d)
    for i, nm in enumerate(('fc_q', 'fc_k', 'fc_v')):
        W, bias = sd[pa + nm + '.weight'], sd[pa + nm + '.bias']
        ref = x0d.double() @ W.double().T + bias.double()
        cpu = (x0d @ W.T + bias).double()
        devq = b['enc0.qkv'].float().cpu().reshape(-1, 3 * d)[:, i * d:(i + 1) * d].double()
        sc = ref.abs().max().item()
        print('%s: max|ref|=%.4g  device rel=%.3g (rms %.3g)   CPU sgemm rel=%.3g (rms %.3g)' % (
            nm, sc, (devq - ref).abs().max().item() / sc, (devq - ref).pow(2).mean().sqrt().item() / sc,
            (cpu - ref).abs().max().item() / sc, (cpu - ref).pow(2).mean().sqrt().item() / sc))
