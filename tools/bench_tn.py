"""TN GEMM (weight gradients) per shape: launch time (main + reduce kernels together), algorithmic TB/s, error against an fp64 reference on a
subsample.  usage: python tools/bench_tn.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
import torch
from hftt_hip import ops

dev = torch.device('cuda:0')
BF = torch.bfloat16
g = torch.Generator().manual_seed(3)
for M in (262144, 90112):
    for (N, K) in ((256, 256), (512, 256), (768, 256), (256, 512)):
        dY = torch.randn(M, N, generator=g).to(dev).to(BF); X = torch.randn(M, K, generator=g).to(dev).to(BF)
        dW, db = ops.gemm_tn(dY, X, npass=1); torch.cuda.synchronize()
        t = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): dW, db = ops.gemm_tn(dY, X, npass=1)
            e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) / 5 * 1e3)
        us = min(t)
        ref = dY[:, :64].double().T @ X.double()
        err = ((dW[:64].double() - ref).abs().max() / ref.abs().max()).item()
        berr = ((db.double() - dY.double().sum(0)).abs().max() / dY.double().sum(0).abs().max()).item()
        print(f'M={M:7d} N={N} K={K}: {us:7.1f} us  {M * (N + K) * 2 / us / 1e6:5.2f} TB/s  rel err dW {err:.1e} db {berr:.1e}', flush=True)
