#!/usr/bin/env python3
"""Where does a strip_linear workgroup spend its life?  Ablation build with bit 64: per-wave shader-clock stamps (QKV shape)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
from hftt_hip import ops, _capi
from hftt_hip._capi import StripDesc, SL_X_BF16, SL_C_BF16, check, lib
dev = torch.device('cuda:0')
M, d = 262144, 256
g = torch.Generator().manual_seed(1)
x = torch.randn(M, d, generator=g).to(dev).to(torch.bfloat16)
W = (torch.randn(3 * d, d, generator=g) / 16).to(dev); b = torch.randn(3 * d, generator=g).to(dev)
wp = ops.strip_pack(W)
out = torch.empty(M, 3 * d, device=dev, dtype=torch.bfloat16)
stamps = torch.zeros(M // 128 * 4 * 16, dtype=torch.int64, device=dev)
dsc = StripDesc()
dsc.M, dsc.N, dsc.K, dsc.flags = M, 3 * d, d, SL_X_BF16 | SL_C_BF16
dsc.x, dsc.ldx, dsc.w, dsc.bias, dsc.C, dsc.ldc, dsc.out_scale = x.data_ptr(), d, wp.data_ptr(), b.data_ptr(), out.data_ptr(), 3 * d, 1.0
dsc.ln_mean = stamps.data_ptr()
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    check(lib().hftt_strip_linear(C.byref(dsc), st), 'strip_linear')
torch.cuda.synchronize()
t = stamps.view(-1, 16).cpu().double()
t0 = t[:, 0].min()
names = ['entry', 'prologue done', 'pass0 mfma end', 'pass0 epi end', 'pass1 mfma end', 'pass1 epi end', 'pass2 mfma end', 'pass2 epi end', 'first slot landed']
order = [0, 1, 8, 2, 3, 4, 5, 6, 7]
print('waves:', t.shape[0], ' kernel span (cycles of s_memtime): %.0f' % (t[:, 7].max() - t0))
prev = None
for k in order:
    col = t[:, k]
    rel = col - t[:, 0]
    msg = '%-18s since wave entry: mean %8.0f  p10 %8.0f  p90 %8.0f' % (names[k], rel.mean(), rel.quantile(0.1), rel.quantile(0.9))
    if prev is not None:
        dlt = col - t[:, prev]
        msg += '   | phase: mean %8.0f' % dlt.mean()
    print(msg)
    prev = k
life = t[:, 7] - t[:, 0]
print('wave life mean %.0f; entry times: generation structure (quantiles of entry - t0): %s' % (life.mean(), [round(float((t[:, 0] - t0).quantile(q))) for q in (0.1, 0.25, 0.4, 0.5, 0.6, 0.75, 0.9)]))
