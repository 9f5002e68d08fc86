#!/usr/bin/env python3
"""dev: the MINI model trained in bf16 mode with dropout 0.1 went NaN after ~1100 steps (tests/test_convergence_gpu.py): find the first
non-finite quantity."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'nylon-amt_amd'), os.path.join(ROOT, 'tests')]
import torch
import util
from util import MINI
import importlib.util
spec = importlib.util.spec_from_file_location('tc', os.path.join(ROOT, 'tests', 'test_convergence_gpu.py')); tc = importlib.util.module_from_spec(spec); spec.loader.exec_module(tc)
from hftt_hip.trainer import TrainStep
dev = torch.device('cuda:0')
cfg, B = MINI, 4
data = tc.make_clips(cfg, 64, 1)
model = util.build_model(cfg, 2025, dropout=0.1).to(dev)
model.hftt_precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
model.train()
ts = TrainStep(model, lr=1e-3)
spec_, labels = data
eng = ts.engine
for s in range(1500):
    idx = [(s * B + i) % 64 for i in range(B)]
    prev = eng.flat_params.clone()
    loss = ts(spec_[idx].to(dev), *[t[idx].to(dev).contiguous() for t in labels])
    l = float(loss[0])
    if not (l == l) or not torch.isfinite(eng.flat_params).all():
        print('step', s, 'loss', loss.cpu().tolist())
        print('params finite before:', bool(torch.isfinite(prev).all()), 'after:', bool(torch.isfinite(eng.flat_params).all()), 'grads finite:', bool(torch.isfinite(eng.flat_grads).all()))
        ws = eng._ws[B]
        for i, o in enumerate(ws['outs']):
            print(' out', i, 'finite', bool(torch.isfinite(o).all()), 'absmax', float(o[torch.isfinite(o)].abs().max()) if torch.isfinite(o).any() else None)
        bad = [(n, int((~torch.isfinite(eng.flat_grads[o:o + k])).sum())) for n, _, o, k in eng._bound if not torch.isfinite(eng.flat_grads[o:o + k]).all()]
        print(' non-finite gradient tensors:', bad[:12], len(bad))
        for name, t in ws['bufs'].items():
            if t.dtype in (torch.float32, torch.bfloat16) and not torch.isfinite(t.float()).all():
                print('  buffer', name, tuple(t.shape), t.dtype, 'non-finite', int((~torch.isfinite(t.float())).sum()))
        b = ws['bufs']
        lse = b['enc0.lse'].view(-1, 2)
        badr = (~torch.isfinite(lse).all(1)).nonzero().flatten().tolist()
        H, L, dh = cfg.enc_head, cfg.n_bin, cfg.hid_dim // cfg.enc_head
        qkv = b['enc0.qkv'].float().view(-1, L, 3, H, dh)
        for r in badr[:3]:
            seq, head, row = r // (H * L), (r // L) % H, r % L
            q = qkv[seq, row, 0, head]; k = qkv[seq, :, 1, head]
            sc = (k @ q)
            print(' bad row', r, 'seq/head/row', seq, head, row, 'lse', lse[r].tolist(), 'raw score max/min', float(sc.max()), float(sc.min()),
                  'q absmax', float(q.abs().max()), 'k absmax', float(k.abs().max()), 'x0 row absmax', float(b['x0'].float().view(-1, L, cfg.hid_dim)[seq, row].abs().max()))
        print(' param absmax before', float(prev.abs().max()))
        big = sorted(((float(prev[o:o + k].abs().max()), n) for n, _, o, k in eng._bound), reverse=True)[:5]
        print(' largest params', big)
        break
    if s % 100 == 0:
        print(s, l, flush=True)
else:
    print('no NaN in 1500 steps')
