#!/usr/bin/env python3
"""dev: per-slot phase times of the pipelined strip linear kernel at the QKV shape (HFTT_STRIP2_DEBUG=4: thread 0 of every workgroup's second
block stamps the shader clock around the phases of the eight slots of pass 1).  EXTRA=1 no fills, EXTRA=2 no barriers."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
os.environ['HFTT_STRIP2_DEBUG'] = str(4 | int(os.environ.get('EXTRA', '0')) | (int(os.environ.get('PASS', '1')) << 8))
print('stamped pass:', os.environ.get('PASS', '1'))
from hftt_hip import ops
from hftt_hip._capi import StripDesc, SL_X_BF16, SL_C_BF16, check, lib
dev = torch.device('cuda:0')
BF = torch.bfloat16
M, d = 262144, 256
N = int(os.environ.get('N', 768))
g = torch.Generator().manual_seed(1)
x = torch.randn(M, d, generator=g).to(dev).to(BF)
W = (torch.randn(N, d, generator=g) / 16).to(dev); b = torch.randn(N, generator=g).to(dev)
wp = ops.strip_pack(W)
y = torch.empty(M, N, device=dev, dtype=BF)
stamps = torch.zeros(2 * 256 * 40, dtype=torch.int64, device=dev)
dsc = StripDesc()
dsc.M, dsc.N, dsc.K, dsc.flags = M, N, d, SL_X_BF16 | SL_C_BF16
dsc.x, dsc.ldx, dsc.w, dsc.bias, dsc.C, dsc.ldc, dsc.out_scale = x.data_ptr(), d, wp.data_ptr(), b.data_ptr(), y.data_ptr(), N, 1.0
dsc.ln_mean = stamps.data_ptr()
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    check(lib().hftt_strip_linear(C.byref(dsc), st), 'strip_linear')
torch.cuda.synchronize()
v3 = os.environ.get('HFTT_STRIP_V3', '1') != '0'
for which in ((0, 1) if v3 else (0,)):
  print('--- wave %d (%s)' % (2 * which, ('filler' if which == 0 else 'fetcher') if v3 else 'second form'))
  t = stamps.view(2, 256, 40)[which].cpu().double()
  sl = t[:, :32].view(256, 8, 4)
  if v3:
    sl = torch.cat([sl[:, :, :3], sl[:, :, 2:3]], dim=2)
  names = ['wait + barrier', '6 reads + 16 MFMA issued', 'refill + x prefetch / stores']
  for k, nm in enumerate(names):
      dlt = (sl[:, :, k + 1] - sl[:, :, k])
      print('%-32s mean %7.0f  p10 %7.0f  p90 %7.0f cycles   per slot: %s' % (nm, dlt.mean(), dlt.reshape(-1).quantile(0.1), dlt.reshape(-1).quantile(0.9),
                                                                                ' '.join('%5.0f' % v for v in dlt.mean(0))))
  gap = sl[:, 1:, 0] - sl[:, :-1, 3]
  print('%-32s mean %7.0f' % ('slot end -> next slot start', gap.mean()))
  print('slot start -> next slot start: mean %.0f cycles' % (sl[:, 1:, 0] - sl[:, :-1, 0]).mean())
  print('pass: 8 slots %.0f cycles, epilogue %.0f cycles' % ((t[:, 32] - sl[:, 0, 0]).mean(), (t[:, 33] - t[:, 32]).mean()))
if not v3:
  print('second block: top of loop -> first slot %.0f ticks (activations of the block copied in: waits for the prefetch), whole block %.0f ticks' % ((t[:, 37] - t[:, 36]).mean(), (t[:, 38] - t[:, 36]).mean()))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10):
    check(lib().hftt_strip_linear(C.byref(dsc), st), 'strip_linear')
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 100
t = stamps.view(2, 256, 40)[0].cpu().double()
print('kernel: %.1f us by events; %.0f ticks from a workgroup\'s first to its last instruction (mean) -> %.2f ticks per ns' % (us, (t[:, 35] - t[:, 34]).mean(), (t[:, 35] - t[:, 34]).mean() / (us * 1e3)))
