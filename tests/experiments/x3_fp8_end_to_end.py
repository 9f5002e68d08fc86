#!/usr/bin/env python3
"""Test infrastructure, not product (it runs the CPU oracle): the arithmetic of DESIGN section 8 (1a) END TO END on the trained paper-size
model.  Every matrix product of the oracle's forward (the Linear layers, Q.K^T, P.V) is replaced by an emulation of
  three : x_hi.W_hi + x_hi.W_lo + x_lo.W_hi on fp16 pairs, fp32 accumulation            (today's x3 mode)
  fp8x  : x_hi.W_hi as fp16, the two cross terms with BOTH factors as e4m3 with a power-of-two scale per 32 elements along k
  fp8x_ffn : fp8x in the two Linear layers of every feed-forward block, three elsewhere
  hi    : x_hi.W_hi alone                                                                 (one pass)
and the nine outputs are compared with the fp32 oracle on the two golden clips of config 5 (tests/golden/config5_paper_*.npz).
Usage: python tests/experiments/x3_fp8_end_to_end.py [clips]      (CPU, a few minutes)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tools', 'experiments'))
import util
from util import O
from pack_checkpoint import unpack_state_dict
from x3_fp8_cross_terms import split_f16, mx_e4m3
import oracle.hftt_oracle as OM

MODE = ['fp32']


def emu_matmul(a, b):
    """a [..., M, K] @ b [..., K, N] in the arithmetic MODE[0] (K padded to a multiple of 32 with zeros for the block scales)"""
    if MODE[0] == 'fp32':
        return _matmul(a, b)
    if MODE[0] == 'fp8x_ffn':                                    # (attention products: three passes)
        MODE[0] = 'three'
        try:
            return emu_matmul(a, b)
        finally:
            MODE[0] = 'fp8x_ffn'
    A = a.detach().numpy().astype(np.float32); B = b.detach().numpy().astype(np.float32)
    K = A.shape[-1]
    pad = (-K) % 32
    if pad:
        A = np.concatenate([A, np.zeros(A.shape[:-1] + (pad,), np.float32)], -1)
        B = np.concatenate([B, np.zeros(B.shape[:-2] + (pad, B.shape[-1]), np.float32)], -2)
    ah, al = split_f16(A); bh, bl = split_f16(B)
    out = np.matmul(ah, bh)
    if MODE[0] == 'three':
        out = out + np.matmul(ah, bl) + np.matmul(al, bh)
    elif MODE[0] == 'fp8x':
        out = out + np.matmul(mx_e4m3(ah, -1), mx_e4m3(bl, -2)) + np.matmul(mx_e4m3(al, -1), mx_e4m3(bh, -2))
    return torch.from_numpy(out.astype(np.float32))


_matmul = torch.matmul
_linear = OM.F.linear


def emu_linear(x, w, b=None):
    if MODE[0] == 'fp8x_ffn':                                    # the fp8 cross terms in the feed-forward pair only, three passes elsewhere
        MODE[0] = 'fp8x' if tuple(w.shape) in ((512, 256), (256, 512)) else 'three'
        try:
            y = emu_matmul(x, w.t())
        finally:
            MODE[0] = 'fp8x_ffn'
    else:
        y = emu_matmul(x, w.t())
    return y if b is None else y + b


def main():
    n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    g = util.golden('config5_paper_golden')
    sd = unpack_state_dict(np.load(os.path.join(util.GOLDEN, 'config5_paper_trained.npz')))
    x = torch.from_numpy(g['input'])[:n_clips]
    torch.set_num_threads(min(8, torch.get_num_threads()))
    outs = {}
    for mode in ('fp32', 'three', 'fp8x', 'fp8x_ffn', 'hi'):
        MODE[0] = mode
        OM.F.linear = emu_linear; OM.torch.matmul = emu_matmul
        try:
            with torch.no_grad():
                outs[mode] = [t.clone() for t in O.model_forward(sd, x, O.PAPER)]
        finally:
            OM.F.linear = _linear; OM.torch.matmul = _matmul
        print('ran', mode, flush=True)
    names = util.OUT_NAMES
    print('%-9s %22s %22s %18s' % ('mode', 'max |posterior diff|', 'max |velocity logit diff|', 'frame decisions'))
    ref = outs['fp32']
    for mode in ('three', 'fp8x', 'fp8x_ffn', 'hi'):
        post = max(float((a - b).abs().max()) for n, a, b in zip(names, outs[mode], ref) if 'velocity' not in n and n != 'attention')
        vel = max(float((a - b).abs().max()) for n, a, b in zip(names, outs[mode], ref) if 'velocity' in n)
        flips = sum(int(((a >= 0.5) != (b >= 0.5)).sum()) for n, a, b in zip(names, outs[mode], ref) if 'velocity' not in n and n != 'attention')
        total = sum(a.numel() for n, a in zip(names, ref) if 'velocity' not in n and n != 'attention')
        print('%-9s %22.3e %22.3e %10d of %d' % (mode, post, vel, flips, total))


if __name__ == '__main__':
    main()
