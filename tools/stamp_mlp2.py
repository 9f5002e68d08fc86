#!/usr/bin/env python3
"""dev: per-slot phase times of the pipelined fused FFN (HFTT_STRIP2_DEBUG=4: stamps of tiles 4 and 5 of each workgroup's second block)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
os.environ['HFTT_STRIP2_DEBUG'] = str(4 | int(os.environ.get('EXTRA', '0')))
from hftt_hip import ops
from hftt_hip._capi import FfnDesc, SL_X_BF16, SL_C_BF16, SL_RES_BF16, check, lib
dev = torch.device('cuda:0')
BF = torch.bfloat16
M, d, pf = 262144, 256, 512
g = torch.Generator().manual_seed(1)
x = torch.randn(M, d, generator=g).to(dev).to(BF)
W1 = (torch.randn(pf, d, generator=g) / 16).to(dev); b1 = torch.randn(pf, generator=g).to(dev)
W2 = (torch.randn(d, pf, generator=g) / 22).to(dev); b2 = torch.randn(d, generator=g).to(dev)
gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev)
wf = ops.ffn_pack(W1, W2)
y = torch.empty(M, d, device=dev, dtype=BF)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
stamps = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
dsc = FfnDesc()
dsc.M, dsc.d, dsc.p, dsc.flags, dsc.mode = M, d, pf, SL_X_BF16 | SL_C_BF16 | SL_RES_BF16, 0
dsc.x, dsc.ldx, dsc.w, dsc.b1, dsc.b2 = x.data_ptr(), d, wf.data_ptr(), b1.data_ptr(), b2.data_ptr()
dsc.ln_gamma, dsc.ln_beta, dsc.ln_mean, dsc.ln_rstd = gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr()
dsc.y, dsc.ldy = y.data_ptr(), d
dsc.gate = stamps.data_ptr()
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    check(lib().hftt_ffn_res_ln_fwd(C.byref(dsc), st), 'ffn')
torch.cuda.synchronize()
t = stamps.view(256, 2, 8).cpu().double()
names = ['slot A: wait + barrier', 'slot A: 16 MFMA + refill + prefetch', 'middle epilogue', 'slot B: wait + barrier', 'slot B: 16 MFMA + refill + stores']
for k, nm in enumerate(names):
    dlt = (t[:, :, k + 1] - t[:, :, k]).reshape(-1)
    print('%-40s mean %7.0f  p10 %7.0f  p90 %7.0f cycles' % (nm, dlt.mean(), dlt.quantile(0.1), dlt.quantile(0.9)))
print('tile 4 start -> tile 5 start: mean %.0f cycles' % (t[:, 1, 0] - t[:, 0, 0]).mean())
