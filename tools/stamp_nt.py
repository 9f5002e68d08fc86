#!/usr/bin/env python3
"""Dev tool (GPU box): in-kernel phase stamps of the one-shot NT GEMM (needs the DEVSTAMP build of gemm_nt.hip)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import torch, numpy as np
from hftt_hip import ops
from hftt_hip._capi import GemmNtDesc, lib, check
dev = torch.device('cuda:0')
M, N, K = 262144, int(sys.argv[1]) if len(sys.argv) > 1 else 768, 256
A = torch.randn(M, K, device=dev)
if len(sys.argv) > 2 and sys.argv[2] == 'abf':
    A = A.to(torch.bfloat16)
W = torch.randn(N, K, device=dev) / 16
b = torch.randn(N, device=dev)
P = ops.prepare_weight(W, 1)
Cout = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
nblk = M // 64
st = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
d = GemmNtDesc()
d.M, d.N, d.K, d.npass = M, N, K, 1
d.A, d.lda, d.W = A.data_ptr(), A.stride(0), P.data_ptr()
d.io_flags = (1 if A.dtype == torch.bfloat16 else 0) | 2
d.bias, d.C, d.ldc, d.act, d.out_scale = b.data_ptr(), Cout.data_ptr(), N, 0, 1.0
d.ln_rstd = st.data_ptr()
s = torch.cuda.current_stream().cuda_stream
for dbg in ((0, 0, 0) if os.environ.get('PLAIN') else (0, 0, 2)):
    d.debug = dbg
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); check(lib().hftt_gemm_nt(C.byref(d), s)); e1.record(); torch.cuda.synchronize()
    print('debug', dbg, 'us', e0.elapsed_time(e1) * 1e3)
if os.environ.get('PLAIN'):
    sys.exit(0)
t = st.cpu().numpy().reshape(nblk, 16)
nst = int((t[0] != 0).sum())
t = t[:, :nst]
dt = np.diff(t, axis=1)
print('stamps per block', nst, ' (start, A staged, [k-loop end, epilogue end] per N tile, drained)')
print('mean cycles per phase:', dt.mean(0).round(0))
print('median:', np.median(dt, 0).round(0))
print('block total mean', (t[:, -1] - t[:, 0]).mean(), 'kernel span', t.max() - t.min())
order = np.argsort(t[:, 0])
print('start times of first 520 blocks (sorted) deciles:', (t[order[:520], 0] - t.min())[::52])
print('blocks 2000..2010 phases:\n', dt[2000:2010])
