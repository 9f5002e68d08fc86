#!/usr/bin/env python3
"""Time the attention kernels alone at the model's shapes (bf16 storage, dropout 0.1): encoder self-attention (1024 sequences x 256 bins,
qkv interleaved [S, 3d]) and the decoder's cross-attention (88 queries over 256 keys, k/v interleaved [S, 2d]).  usage: bench_attn.py [fwd|bwd|both]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'nylon-amt_amd')]
import torch   # noqa: E402
from hftt_hip import ops   # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'both'
dev = torch.device('cuda', 0)
BF = torch.bfloat16
H, d = 4, 256


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


g = torch.Generator(device='cpu').manual_seed(1)
for name, n_seq, Lq, Lk in (('enc self (256x256)', 1024, 256, 256), ('cross (88x256)', 1024, 88, 256), ('time self (128x128)', 704, 128, 128)):
    if Lq == Lk:                    # the engine's layouts: qkv (and their gradients) interleaved per token
        qkv = (torch.randn(n_seq, Lq, 3 * d, generator=g) * 0.5).to(dev).to(BF)
        q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
        gq = torch.empty_like(qkv)
        grads = (gq[:, :, :d], gq[:, :, d:2 * d], gq[:, :, 2 * d:])
    else:
        q = (torch.randn(n_seq, Lq, d, generator=g) * 0.5).to(dev).to(BF)
        kv = (torch.randn(n_seq, Lk, 2 * d, generator=g) * 0.5).to(dev).to(BF)
        k, v = kv[:, :, :d], kv[:, :, d:]
        gkv = torch.empty_like(kv)
        grads = (torch.empty_like(q), gkv[:, :, :d], gkv[:, :, d:])
    kw = dict(npass=1, drop_p=0.1, drop_site=3, drop_seed=77)
    out, lse = ops.attn_fwd(q, k, v, H, out_dtype=BF, **kw)
    dout = (torch.randn(n_seq, Lq, d, generator=g) * 0.1).to(dev).to(BF)
    line = '%-22s' % name
    if what in ('fwd', 'both'):
        line += ' fwd %7.1f us' % timed(lambda: ops.attn_fwd(q, k, v, H, out_dtype=BF, **kw))
        line += ' (p=0: %7.1f us)' % timed(lambda: ops.attn_fwd(q, k, v, H, out_dtype=BF, npass=1))
    if what in ('bwd', 'both'):
        line += ' bwd %7.1f us' % timed(lambda: ops.attn_bwd(q, k, v, out, lse, dout, H, grads_out=grads, **kw))
    print(line, flush=True)
