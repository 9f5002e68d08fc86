"""LayerNorm backward at the encoder's shape (262,144 x 256) in the x3 plan's storage mix (fp32 dy and dr, bf16 saved sum, bf16 dropped copy)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
from hftt_hip import ops
dev = torch.device('cuda:0')
M, N = int(os.environ.get('M', 262144)), int(os.environ.get('N', 256))
g = torch.Generator().manual_seed(1)
dy = torch.randn(M, N, generator=g).to(dev); r = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
mean = torch.zeros(M).to(dev); rstd = torch.ones(M).to(dev); gam = torch.randn(N, generator=g).to(dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for p, dd in ((0.0, torch.float32), (0.1, torch.bfloat16)):
    us = t(lambda: ops.ln_bwd(dy, r, mean, rstd, gam, drop_p=p, drop_site=1, drop_seed=2, drop_dtype=dd))
    byts = M * N * (4 + 2 + 4 + (2 if p > 0 else 0))
    print('variant %s drop %.1f: %.1f us  %.2f TB/s' % (os.environ.get('HFTT_LNB_VARIANT', '0'), p, us, byts / us / 1e6), flush=True)
