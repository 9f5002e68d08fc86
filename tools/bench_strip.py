#!/usr/bin/env python3
"""Time the strip kernels at the paper-size shapes of one training step (B = 8: 262,144 bin tokens) next to the round-1
kernels they replace.  Prints one line per case: microseconds, TFLOP/s, algorithmic GB/s.  Dev tool (GPU box)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
from hftt_hip import ops   # noqa: E402

BF = torch.bfloat16
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
    print('%-44s %8.1f us  %7.1f TFLOP/s  %7.1f GB/s' % (name, us, flops / us / 1e6, nbytes / us / 1e3), flush=True)


def main():
    only_strip = len(sys.argv) > 1 and sys.argv[1] in ('strip', 'ffn')
    ffn_only = len(sys.argv) > 1 and sys.argv[1] == 'ffn'
    M = int(os.environ.get('M', 262144))
    d, p = 256, 512
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, d, generator=g).to(dev)
    xb = x.to(BF)
    Wq = (torch.randn(3 * d, d, generator=g) / 16).to(dev); bq = torch.randn(3 * d, generator=g).to(dev)
    Wo = (torch.randn(d, d, generator=g) / 16).to(dev); bo = torch.randn(d, generator=g).to(dev)
    W1 = (torch.randn(p, d, generator=g) / 16).to(dev); b1 = torch.randn(p, generator=g).to(dev)
    W2 = (torch.randn(d, p, generator=g) / 22).to(dev); b2 = torch.randn(d, generator=g).to(dev)
    gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev)
    res = torch.randn(M, d, generator=g).to(dev); resb = res.to(BF)

    if ffn_only:
        wf = ops.ffn_pack(W1, W2)
        fl = 4.0 * M * d * p
        line('ffn  fused training (h + pre saved, p=0.1)', timeit(lambda: ops.ffn_res_ln_fwd(xb, wf, p, b1, b2, gam, bet, drop_p=0.1, site_h=4, site_o=5, seed=7)),
             fl, M * d * 2 * 3 + M * p * 2)
        line('ffn  fused inference (nothing saved)', timeit(lambda: ops.ffn_res_ln_fwd(xb, wf, p, b1, b2, gam, bet, save_hidden=False, save_pre=False)),
             fl, M * d * 2 * 2)
        return
    # ---- QKV projection ----
    wq_s = ops.strip_pack(Wq); wq_o = ops.prepare_weight(Wq, 1)
    fl = 2.0 * M * 3 * d * d
    only_strip or line('qkv  old (x fp32 -> bf16)', timeit(lambda: ops.gemm_nt(x, Wq, bq, npass=1, planes=wq_o, out_dtype=BF)), fl, M * d * 4 + M * 3 * d * 2)
    only_strip or line('qkv  old (x bf16 -> bf16)', timeit(lambda: ops.gemm_nt(xb, Wq, bq, npass=1, planes=wq_o, out_dtype=BF)), fl, M * d * 2 + M * 3 * d * 2)
    line('qkv  strip (x bf16 -> bf16)', timeit(lambda: ops.strip_linear(xb, wq_s, 3 * d, bias=bq)), fl, M * d * 2 + M * 3 * d * 2)
    only_strip or line('qkv  strip (x fp32 -> bf16)', timeit(lambda: ops.strip_linear(x, wq_s, 3 * d, bias=bq)), fl, M * d * 4 + M * 3 * d * 2)
    # ---- fc_o + dropout + residual + LayerNorm ----
    wo_s = ops.strip_pack(Wo); wo_o = ops.prepare_weight(Wo, 1)
    fl = 2.0 * M * d * d
    only_strip or line('o+LN old (ctx bf16, res/out/pre fp32)', timeit(lambda: ops.gemm_nt(xb, Wo, bo, npass=1, planes=wo_o, drop_p=0.1, drop_site=3, drop_seed=7, residual=res,
                                                                             ln=(gam, bet))), fl, M * d * (2 + 4 + 4 + 4))
    line('o+LN strip (all bf16, pre saved)', timeit(lambda: ops.strip_linear(xb, wo_s, d, bias=bo, drop_p=0.1, drop_site=3, drop_seed=7, residual=resb,
                                                                            ln=(gam, bet))), fl, M * d * 2 * 4)
    line('o+LN strip (all bf16, inference)', timeit(lambda: ops.strip_linear(xb, wo_s, d, bias=bo, residual=resb, ln=(gam, bet), save_pre=False)),
         fl, M * d * 2 * 3)
    # ---- FFN ----
    wf = ops.ffn_pack(W1, W2); w1_o = ops.prepare_weight(W1, 1); w2_o = ops.prepare_weight(W2, 1)
    fl = 4.0 * M * d * p

    def old_ffn():
        h = ops.gemm_nt(x, W1, b1, npass=1, planes=w1_o, act=1, drop_p=0.1, drop_site=4, drop_seed=7, out_dtype=BF)
        return ops.gemm_nt(h, W2, b2, npass=1, planes=w2_o, drop_p=0.1, drop_site=5, drop_seed=7, residual=x, ln=(gam, bet))
    only_strip or line('ffn  old (2 launches, fp32 stream)', timeit(old_ffn), fl, M * d * 4 + M * p * 2 * 2 + M * d * 4 * 3)
    line('ffn  fused training (h + pre saved, p=0.1)', timeit(lambda: ops.ffn_res_ln_fwd(xb, wf, p, b1, b2, gam, bet, drop_p=0.1, site_h=4, site_o=5, seed=7)),
         fl, M * d * 2 * 3 + M * p * 2)
    line('ffn  fused training (p=0)', timeit(lambda: ops.ffn_res_ln_fwd(xb, wf, p, b1, b2, gam, bet)), fl, M * d * 2 * 3 + M * p * 2)
    line('ffn  fused inference (nothing saved)', timeit(lambda: ops.ffn_res_ln_fwd(xb, wf, p, b1, b2, gam, bet, save_hidden=False, save_pre=False)),
         fl, M * d * 2 * 2)
    hid = torch.relu(torch.randn(M, p, generator=g)).to(dev).to(BF)
    wfb = ops.ffn_pack(W1, W2, backward=True)
    line('ffn  fused backward dX (dh saved)', timeit(lambda: ops.ffn_bwd_dx(xb, wfb, p, hid, gate_scale=1.11, residual=resb)), fl,
         M * d * 2 * 3 + M * p * 2 * 2)
    # ---- dX of the QKV projection: K = 768 -> 256, + residual ----
    dq = torch.randn(M, 3 * d, generator=g).to(dev).to(BF)
    wqt_s = ops.strip_pack(Wq, transpose=True); wqt_o = ops.prepare_weight(Wq, 1, transposed=True)
    fl = 2.0 * M * 3 * d * d
    WqT = Wq.T.contiguous()
    only_strip or line('dqkv old (K=768, fp32 res/out)', timeit(lambda: ops.gemm_nt(dq, WqT, None, npass=1, planes=wqt_o, residual=res)), fl, M * 3 * d * 2 + M * d * 8)
    line('dqkv strip (K=768, bf16 res/out)', timeit(lambda: ops.strip_linear(dq, wqt_s, d, residual=resb)), fl, M * 3 * d * 2 + M * d * 4)


if __name__ == '__main__':
    main()
