#!/usr/bin/env python3
"""GPU box: where do the two register plans of the bf16 fused feed-forward block (HFTT_MLP2_WPC=1 / 2) differ?  Same inputs, every output
tensor compared element for element; repeated launches of ONE plan against each other (a race would show as run-to-run differences)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import torch
from hftt_hip import ops
BF = torch.bfloat16
dev = torch.device('cuda:0')


def run(wpc, x, wf, pf, b1, b2, gam, bet, p_, save):
    os.environ['HFTT_MLP2_WPC'] = str(wpc)
    out = ops.ffn_res_ln_fwd(x, wf, pf, b1, b2, gam, bet, drop_p=p_, site_h=4, site_o=5, seed=7, save_hidden=save, save_pre=save)
    torch.cuda.synchronize()
    return out


def main():
    for M in (128, 4096, 140032):
        g = torch.Generator().manual_seed(M + 1)
        d, pf = 256, 512
        x = torch.randn(M, d, generator=g).to(dev).to(BF)
        W1 = (torch.randn(pf, d, generator=g) / 16).to(dev); b1 = (0.5 * torch.randn(pf, generator=g)).to(dev)
        W2 = (torch.randn(d, pf, generator=g) / 22).to(dev); b2 = (0.5 * torch.randn(d, generator=g)).to(dev)
        gam = (1 + 0.3 * torch.randn(d, generator=g)).to(dev); bet = torch.randn(d, generator=g).to(dev)
        wf = ops.ffn_pack(W1, W2)
        for p_, save in ((0.1, True), (0.0, True), (0.0, False)):
            a = run(1, x, wf, pf, b1, b2, gam, bet, p_, save)
            b = run(2, x, wf, pf, b1, b2, gam, bet, p_, save)
            b2_ = run(2, x, wf, pf, b1, b2, gam, bet, p_, save)
            rep = []
            for name, u, v, w in zip(('y', 'hidden', 'pre_ln', 'mean', 'rstd'), a, b, b2_):
                if u is None:
                    continue
                ne = int((u != v).sum()); rr = int((v != w).sum())
                mx = float((u.float() - v.float()).abs().max())
                rep.append('%s: %d of %d differ (max %.3g), run-to-run %d' % (name, ne, u.numel(), mx, rr))
            print('M %6d p %.1f save %d | %s' % (M, p_, save, ' | '.join(rep)), flush=True)


if __name__ == '__main__':
    main()
