"""x3 TN GEMMs (weight gradients, split bf16 operands: npass 4) at the shapes of the paper-size step: launch time (main + reduce kernels together)
and algorithmic TB/s.  usage: python tools/bench_tn_x3.py   (tools/ablate_tn.sh runs it on a build whose loader does not split)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
import torch
from hftt_hip import ops

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
for M in (262144, 90112):
    for (N, K, ybf, xbf) in ((768, 256, False, False), (256, 256, False, False), (512, 256, False, False), (256, 512, False, True), (512, 256, True, False)):
        dY = torch.randn(M, N, generator=g).to(dev); X = torch.randn(M, K, generator=g).to(dev)
        if ybf: dY = dY.bfloat16()
        if xbf: X = X.bfloat16()
        dW, db = ops.gemm_tn(dY, X, npass=4); torch.cuda.synchronize()
        t = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): dW, db = ops.gemm_tn(dY, X, npass=4)
            e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) / 5 * 1e3)
        us = min(t)
        nbytes = M * (N * dY.element_size() + K * X.element_size())
        print(f'M={M:7d} N={N} K={K} dY {"bf16" if ybf else "fp32"} X {"bf16" if xbf else "fp32"}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s', flush=True)
