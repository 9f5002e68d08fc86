#!/usr/bin/env python3
"""dev: run one variant of the pipelined fused FFN kernel and compare with the one-block-per-workgroup form.  usage: dbg_mlp2.py <variant> [M]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
from hftt_hip import ops
BF = torch.bfloat16
dev = torch.device('cuda:0')
variant = sys.argv[1]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator().manual_seed(3)
d, pf = 256, 512
x = torch.randn(M, d, generator=g).to(dev).to(BF)
W1 = (torch.randn(pf, d, generator=g) / 16).to(dev); b1 = (0.5 * torch.randn(pf, generator=g)).to(dev)
W2 = (torch.randn(d, pf, generator=g) / 22).to(dev); b2 = (0.5 * torch.randn(d, generator=g)).to(dev)
gam = (1 + 0.3 * torch.randn(d, generator=g)).to(dev); bet = torch.randn(d, generator=g).to(dev)
wf = ops.ffn_pack(W1, W2); wfb = ops.ffn_pack(W1, W2, backward=True)
hid = torch.relu(torch.randn(M, pf, generator=g)).to(dev).to(BF)
res = torch.randn(M, d, generator=g).to(dev).to(BF)


def run():
    if variant == 'bwd':
        return ops.ffn_bwd_dx(x, wfb, pf, hid, gate_scale=1.0 / 0.9, residual=res)
    if variant == 'bwd_nores':
        return ops.ffn_bwd_dx(x, wfb, pf, hid, gate_scale=1.0)
    if variant == 'inf':
        return ops.ffn_res_ln_fwd(x, wf, pf, b1, b2, gam, bet, save_hidden=False, save_pre=False)
    if variant == 'train0':
        return ops.ffn_res_ln_fwd(x, wf, pf, b1, b2, gam, bet)
    return ops.ffn_res_ln_fwd(x, wf, pf, b1, b2, gam, bet, drop_p=0.1, site_h=4, site_o=5, seed=7)


os.environ['HFTT_STRIP_V2'] = '0'
a = run(); torch.cuda.synchronize()
print(variant, 'v1 done', flush=True)
os.environ['HFTT_STRIP_V2'] = '1'
b = run(); torch.cuda.synchronize()
print(variant, 'v2 done', flush=True)
for u, v in zip(a, b):
    if u is not None:
        print('  equal' if torch.equal(u, v) else '  DIFF max %g' % (u.float() - v.float()).abs().max().item(), tuple(u.shape), flush=True)
