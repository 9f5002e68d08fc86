#!/usr/bin/env python3
"""Time the split-operand (x3) strip kernels at the paper-size shapes of one training step (B = 8: 262,144 bin tokens).  With
HFTT_X3_DEBUG set (csrc/x3_strip.hip) single mechanisms are switched off so their cost can be read from the difference (results are
then garbage).  Prints one line per case: microseconds, algorithmic TFLOP/s (3x that on the matrix pipe), algorithmic GB/s.  Dev tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
from hftt_hip import ops   # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def line(name, us, flops, nbytes):
    print('%-40s %8.1f us  %7.1f TFLOP/s  %7.1f GB/s   (debug %s)' % (name, us, flops / us / 1e6, nbytes / us / 1e3, os.environ.get('HFTT_X3_DEBUG', '0')), flush=True)


def main():
    M = int(os.environ.get('M', 262144))
    d, p = 256, 512
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, d, generator=g).to(dev)
    x3 = torch.randn(M, 3 * d, generator=g).to(dev)
    Wq = (torch.randn(3 * d, d, generator=g) / 16).to(dev); bq = torch.randn(3 * d, generator=g).to(dev)
    Wo = (torch.randn(d, d, generator=g) / 16).to(dev); bo = torch.randn(d, generator=g).to(dev)
    W1 = (torch.randn(p, d, generator=g) / 16).to(dev); b1 = torch.randn(p, generator=g).to(dev)
    W2 = (torch.randn(d, p, generator=g) / 22).to(dev); b2 = torch.randn(d, generator=g).to(dev)
    gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev)
    res = torch.randn(M, d, generator=g).to(dev)
    ffn_only = os.environ.get('FFN_ONLY') == '1'
    qkv_only = os.environ.get('QKV_ONLY') == '1'
    wq = ops.x3_strip_pack(Wq, 2, order=1)
    ffn_only or line('qkv [M,256]->[M,768]', timeit(lambda: ops.strip_linear(x, wq, 3 * d, bias=bq, x3=2)), 2.0 * M * 3 * d * d, 4.0 * M * 4 * d)
    if qkv_only:
        return
    wo = ops.x3_strip_pack(Wo, 2)
    ffn_only or line('o-proj + drop + res + LN', timeit(lambda: ops.strip_linear(x, wo, d, bias=bo, drop_p=0.1, drop_site=3, drop_seed=7, residual=res, ln=(gam, bet), x3=2)),
         2.0 * M * d * d, 4.0 * M * 4 * d)
    ffn_only or line('linear 256->256 plain', timeit(lambda: ops.strip_linear(x, wo, d, bias=bo, x3=2)), 2.0 * M * d * d, 4.0 * M * 2 * d)
    wqt = ops.x3_strip_pack(Wq, 4, transpose=True)
    ffn_only or line('dX K=768 + residual', timeit(lambda: ops.strip_linear(x3, wqt, d, residual=res, x3=4)), 2.0 * M * 3 * d * d, 4.0 * M * 5 * d)
    wf = ops.x3_ffn_pack(W1, W2)
    line('ffn fwd (training: saves h, pre)', timeit(lambda: ops.ffn_res_ln_fwd(x, wf, p, b1, b2, gam, bet, drop_p=0.1, site_h=1, site_o=2, seed=5, x3=True)),
         4.0 * M * d * p, 4.0 * M * (3 * d + p))
    line('ffn fwd (inference)', timeit(lambda: ops.ffn_res_ln_fwd(x, wf, p, b1, b2, gam, bet, save_hidden=False, save_pre=False, x3=True)),
         4.0 * M * d * p, 4.0 * M * 2 * d)
    line('ffn fwd (training, hidden as bf16)', timeit(lambda: ops.ffn_res_ln_fwd(x, wf, p, b1, b2, gam, bet, drop_p=0.1, site_h=1, site_o=2, seed=5, x3=True, hidden_bf16=True)),
         4.0 * M * d * p, 4.0 * M * 3 * d + 2.0 * M * p)
    hid = torch.relu(torch.randn(M, p, generator=g)).to(dev)
    wb = ops.x3_ffn_pack(W1, W2, backward=True)
    line('ffn bwd dx', timeit(lambda: ops.ffn_bwd_dx(x, wb, p, hid, gate_scale=1.1, residual=res, x3=True)), 4.0 * M * d * p, 4.0 * M * (3 * d + 2 * p))
    hb = hid.bfloat16()
    line('ffn bwd dx (hidden / dh as bf16)', timeit(lambda: ops.ffn_bwd_dx(x, wb, p, hb, gate_scale=1.1, residual=res, x3=True)), 4.0 * M * d * p, 4.0 * M * 3 * d + 2.0 * M * 2 * p)


if __name__ == '__main__':
    main()
