#!/usr/bin/env python3
"""Dev tool (GPU box): time single kernels of the path at paper-size shapes (B=8) through the C ABI.
Usage: python tools/bench_kernels.py [filter] [--iters N]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import torch
from hftt_hip import ops

dev = torch.device('cuda:0')
flt = [a for a in sys.argv[1:] if not a.startswith('--')]
iters = 20
for i, a in enumerate(sys.argv):
    if a == '--iters': iters = int(sys.argv[i + 1])
Se, Sn, d, p = 8 * 128 * 256, 8 * 128 * 88, 256, 512


def timeit(name, fn, nbytes, flops):
    if flt and not any(f in name for f in flt):
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print('%-34s %9.1f us  %7.0f GB/s  %7.1f TF/s' % (name, us, nbytes / us / 1e3, flops / us / 1e6))


g = torch.Generator(device='cpu').manual_seed(0)
x = torch.randn(Se, d, device=dev)
y = torch.empty_like(x)
timeit('copy 268MB (torch)', lambda: y.copy_(x), 2 * x.numel() * 4, 0)

for npass in (1,):
    W768 = torch.randn(768, 256, device=dev) / 16
    W256 = torch.randn(256, 256, device=dev) / 16
    W512 = torch.randn(512, 256, device=dev) / 16
    W2 = torch.randn(256, 512, device=dev) / 22
    b768 = torch.randn(768, device=dev); b256 = torch.randn(256, device=dev); b512 = torch.randn(512, device=dev)
    gam = torch.ones(256, device=dev); bet = torch.zeros(256, device=dev)
    h = torch.randn(Se, p, device=dev)
    P768 = ops.prepare_weight(W768, npass); P256 = ops.prepare_weight(W256, npass); P512 = ops.prepare_weight(W512, npass); P2 = ops.prepare_weight(W2, npass)
    timeit('nt qkv   M=262144 N=768 K=256', lambda: ops.gemm_nt(x, W768, b768, npass=npass, planes=P768), 4 * Se * (256 + 768), 2.0 * Se * 768 * 256)
    timeit('nt plain M=262144 N=256 K=256', lambda: ops.gemm_nt(x, W256, b256, npass=npass, planes=P256), 4 * Se * 512, 2.0 * Se * 256 * 256)
    timeit('nt o+res+LN      N=256 K=256', lambda: ops.gemm_nt(x, W256, b256, npass=npass, planes=P256, residual=y, ln=(gam, bet)), 4 * Se * 256 * 4, 2.0 * Se * 256 * 256)
    timeit('nt f1 relu drop  N=512 K=256', lambda: ops.gemm_nt(x, W512, b512, npass=npass, planes=P512, act=1, drop_p=0.1, drop_site=1, drop_seed=5), 4 * Se * (256 + 512), 2.0 * Se * 512 * 256)
    timeit('nt f2+res+LN     N=256 K=512', lambda: ops.gemm_nt(h, W2, b256, npass=npass, planes=P2, residual=y, ln=(gam, bet)), 4 * Se * (512 + 768), 2.0 * Se * 256 * 512)
    W3 = torch.randn(256, 768, device=dev) / 27; P3 = ops.prepare_weight(W3, npass); g3 = torch.randn(Se, 768, device=dev)
    timeit('nt dX qkv_t+res  N=256 K=768', lambda: ops.gemm_nt(g3, W3, None, npass=npass, planes=P3, residual=y), 4 * Se * (768 + 512), 2.0 * Se * 256 * 768)
    timeit('tn dW   M=262144 N=256 K=256', lambda: ops.gemm_tn(x, y, npass=npass), 4 * Se * 512, 2.0 * Se * 256 * 256)
    timeit('tn dW   M=262144 N=512 K=256', lambda: ops.gemm_tn(h, y, npass=npass), 4 * Se * 768, 2.0 * Se * 512 * 256)
    qkv = torch.randn(1024, 256, 768, device=dev)
    q, k, v = qkv[..., :256], qkv[..., 256:512], qkv[..., 512:]
    out, lse = ops.attn_fwd(q, k, v, 4, npass=npass)
    do = torch.randn_like(out)
    timeit('attn fwd 1024x4 256x256x64', lambda: ops.attn_fwd(q, k, v, 4, npass=npass), 4 * Se * 256 * 4, 4.0 * 1024 * 4 * 256 * 256 * 64)
    timeit('attn bwd 1024x4 256x256x64', lambda: ops.attn_bwd(q, k, v, out, lse, do, 4, npass=npass), 4 * Se * 256 * 8, 10.0 * 1024 * 4 * 256 * 256 * 64)
    mean = torch.zeros(Se, device=dev); rstd = torch.ones(Se, device=dev)
    timeit('ln_bwd M=262144 N=256', lambda: ops.ln_bwd(x, y, mean, rstd, gam), 4 * Se * 256 * 3, 0)
