#!/usr/bin/env python3
"""dev: bf16-storage attention (HB kernels) with dropout against the fp64 reference with the emulated masks, several geometries"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'nylon-amt_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
from hftt_hip import ops
import util
dev = torch.device('cuda:0'); BF = torch.bfloat16
for (n, H, Lq, Lk, dh) in ((6, 4, 8, 8, 64), (5, 4, 16, 16, 64), (4, 4, 8, 32, 64), (3, 4, 88, 256, 64), (3, 2, 256, 256, 64), (2, 4, 128, 128, 64), (3, 4, 88, 88, 64)):
    g = torch.Generator().manual_seed(Lq * 1000 + Lk)
    d = H * dh
    q = (torch.randn(n, Lq, d, generator=g) * 0.3).to(BF); k = (torch.randn(n, Lk, d, generator=g) * 0.3).to(BF); v = torch.randn(n, Lk, d, generator=g).to(BF)
    p, site, seed = 0.25, 7, 4242
    out, lse = ops.attn_fwd(q.to(dev), k.to(dev), v.to(dev), H, npass=1, drop_p=p, drop_site=site, drop_seed=seed, out_dtype=BF)
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    def heads(t, L): return t.view(n, L, H, dh).permute(0, 2, 1, 3)
    s = heads(q64, Lq) @ heads(k64, Lk).transpose(-1, -2) / math.sqrt(dh)
    P = torch.softmax(s, -1)
    mask = util.keep_mask_t(seed, site, (n, H, Lq, Lk), p).double()
    o = ((P * mask / (1 - p)) @ heads(v64, Lk)).permute(0, 2, 1, 3).reshape(n, Lq, d)
    e_f = util.rel_err(out, o)
    do = torch.randn(n, Lq, d, generator=g).to(BF)
    (o * do.double()).sum().backward()
    dq, dk, dv = ops.attn_bwd(q.to(dev), k.to(dev), v.to(dev), out, lse, do.to(dev), H, npass=1, drop_p=p, drop_site=site, drop_seed=seed, dq_dtype=BF, dkv_dtype=BF)
    print('Lq %3d Lk %3d: fwd rel err %.4f   dq %.4f dk %.4f dv %.4f' % (Lq, Lk, e_f, util.rel_err(dq, q64.grad), util.rel_err(dk, k64.grad), util.rel_err(dv, v64.grad)), flush=True)
