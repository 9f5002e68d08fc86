#!/usr/bin/env python3
"""dev: phase times of one query block of the x3 attention backward on planes (-DHFTT_X3_ATTN_STAMPS build: tools/stamp_x3_attn.sh)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
from hftt_hip import ops
from hftt_hip.ops import _attn_desc, _stream
from hftt_hip._capi import check, lib
dev = torch.device('cuda:0')
n, H, dh, Lq, Lk = 1024, 4, 64, int(os.environ.get('LQ', 256)), 256
d = H * dh
g = torch.Generator().manual_seed(1)
q = torch.randn(n, Lq, d, generator=g).to(dev); kv = torch.randn(n, Lk, 2 * d, generator=g).to(dev); do = torch.randn(n, Lq, d, generator=g).to(dev)
qp, kvp = ops.to_planes(q), ops.to_planes(kv)
kp, vp = kvp[..., :d], kvp[..., d:]
out, lse = ops.attn_fwd(q, kv[..., :d], kv[..., d:], H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3)
dq = torch.empty_like(q); dkv = torch.empty_like(kv)
stamps = torch.zeros(64 * 8 * 32, dtype=torch.int64, device=dev)
dsc = _attn_desc(qp, kp, vp, H, 2, 0.1, 1, 3, True)
dsc.out, dsc.o_seq_stride, dsc.ldo = out.data_ptr(), out.stride(0), out.stride(1)
dsc.lse, dsc.dout = lse.data_ptr(), do.data_ptr()
dsc.dq, dsc.dq_seq_stride, dsc.lddq = dq.data_ptr(), dq.stride(0), dq.stride(1)
dk, dv = dkv[..., :d], dkv[..., d:]
dsc.dk, dsc.dk_seq_stride, dsc.lddk = dk.data_ptr(), dk.stride(0), dk.stride(1)
dsc.dv, dsc.dv_seq_stride, dsc.lddv = dv.data_ptr(), dv.stride(0), dv.stride(1)
dsc.probs = stamps.data_ptr()
for _ in range(int(os.environ.get('WARM', 3))):      # (WARM=300: a third of a second of the same launches first -- the first kernels of a process are clocked lower)
    check(lib().hftt_attn_bwd(C.byref(dsc), _stream(dev)), 'attn_bwd')
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(8): check(lib().hftt_attn_bwd(C.byref(dsc), _stream(dev)), 'attn_bwd')
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 8 * 1e3
print('launch %.1f us' % us)
t = stamps.view(64, 8, 32).cpu().double()
names = ['(a) staging: registers -> LDS', 'barrier (b)', '(c) S / dP: 24 MFMA', '(d) softmax backward (+ dQ steps when interleaved)', 'padding, prefetch, dQ store',
         '(e) splits + dS -> LDS', '(f) dV / dK: 24 MFMA', 'barrier (h)', '(i) dQ phase / loop end']
for k, nm in enumerate(names):
    dl = (t[:, :, k + 1] - t[:, :, k]).reshape(-1)
    print('%-52s mean %7.0f  p10 %7.0f  p90 %7.0f' % (nm, dl.mean(), dl.quantile(0.1), dl.quantile(0.9)))
print('%-52s mean %7.0f' % ('whole block', (t[:, :, 9] - t[:, :, 0]).mean()))
for nm, a_, b_ in (('prologue: K image, K / V fragments, first prefetch', 10, 11), ('the query-block loop', 11, 12), ('last dQ (interleaved form)', 12, 13),
                   ('epilogue: dK / dV through LDS, stores drained', 13, 14), ('whole (sequence, head) item', 10, 14)):
    dl = (t[:, :, b_] - t[:, :, a_]).reshape(-1)
    print('%-52s mean %7.0f  p10 %7.0f  p90 %7.0f' % (nm, dl.mean(), dl.quantile(0.1), dl.quantile(0.9)))
print('shader clock over the stamped items: %.2f GHz (s_memtime ticks per 100 MHz s_memrealtime tick)' % (((t[:, :, 14] - t[:, :, 10]) / (t[:, :, 17] - t[:, :, 16])).mean() * 0.1))
wg = t[:, 0, 10].sort().values
print('workgroups 1024 .. 1087 start over %.0f ticks; item ticks x 16 items / launch time = %.2f GHz' % (wg[-1] - wg[0], (t[:, :, 14] - t[:, :, 10]).mean() * 16 / (us * 1e3)))
