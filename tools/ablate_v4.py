"""Timing experiments on the mover-wave strip kernel (HFTT_STRIP4_DEBUG bits; results are garbage under any bit)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
import torch
from hftt_hip import ops

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
M = 262144
shapes = [(768, 256)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (N, K) in shapes:
    x = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) / 16).to(dev); b = torch.randn(N, generator=g).to(dev)
    w = ops.strip_pack(W)
    os.environ['HFTT_STRIP_V4'] = '1'
    for bits, what in ((0, 'full'), (1, 'no stores'), (2, 'no x refresh'), (3, 'no stores, no x'), (4, 'no fills'), (7, 'movers idle'), (15, 'movers idle, no staging'),
                       (16 + 15, 'movers idle, no staging, no barriers'), (64, 'compute idle'), (64 + 1, 'compute idle, no stores'), (64 + 2, 'compute idle, no x'),
                       (64 + 4, 'compute idle, no fills'), (64 + 3, 'compute idle, fills only'), (64 + 6, 'compute idle, stores only'), (64 + 16, 'compute idle, no barriers')):
        os.environ['HFTT_STRIP4_DEBUG'] = str(bits)
        y = ops.strip_linear(x, w, N, bias=b); torch.cuda.synchronize()
        t = []
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): y = ops.strip_linear(x, w, N, bias=b)
            e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) / 5 * 1e3)
        print(f'N={N} K={K} bits={bits:3d} {what:28s}: {min(t):7.1f} us', flush=True)
    os.environ['HFTT_STRIP4_DEBUG'] = '0'
