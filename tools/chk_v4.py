"""Mover-wave strip kernel (csrc/strip_gemm4.hip) against the second form: bit identity and launch time per shape."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
import torch
from hftt_hip import ops

dev = torch.device('cuda:0')
BF = torch.bfloat16
g = torch.Generator().manual_seed(5)
ok = True
for M in (128, 4096, 38432, 262144):
    for (N, K) in ((768, 256), (512, 256), (256, 256), (256, 512), (256, 768)):
        x = torch.randn(M, K, generator=g).to(dev).to(BF)
        W = (torch.randn(N, K, generator=g) / 16).to(dev); b = torch.randn(N, generator=g).to(dev)
        w = ops.strip_pack(W)
        outs = {}
        for form in ('0', '1'):
            os.environ['HFTT_STRIP_V4'] = form
            y = ops.strip_linear(x, w, N, bias=b)
            torch.cuda.synchronize()
            t = []
            for _ in range(3):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): y = ops.strip_linear(x, w, N, bias=b)
                e1.record(); torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) / 5 * 1e3)
            outs[form] = (y.clone(), min(t))
        same = torch.equal(outs['0'][0], outs['1'][0])
        ok &= same
        nbad = (outs['0'][0] != outs['1'][0]).sum().item()
        print(f'M={M:7d} N={N} K={K}: v2 {outs["0"][1]:7.1f} us  v4 {outs["1"][1]:7.1f} us  identical={same} (differing {nbad})', flush=True)
print('ALL IDENTICAL' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
