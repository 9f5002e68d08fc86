"""An opt-in strip kernel form (HFTT_STRIP_V4 / HFTT_STRIP_V5) against the default form: bit identity and launch time per shape.
usage: python tools/chk_forms.py V5 [M ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
import torch
from hftt_hip import ops

form_var = 'HFTT_STRIP_' + (sys.argv[1] if len(sys.argv) > 1 else 'V5')
Ms = [int(a) for a in sys.argv[2:]] or [128, 4096, 38432, 90112, 262144]
dev = torch.device('cuda:0')
BF = torch.bfloat16
g = torch.Generator().manual_seed(5)
ok = True
for M in Ms:
    for (N, K, hr) in ((768, 256, False), (512, 256, False), (256, 256, False), (256, 512, False), (256, 768, False), (256, 768, True), (256, 512, True), (256, 256, True)):
        if hr and form_var.endswith('V4'): continue
        x = torch.randn(M, K, generator=g).to(dev).to(BF)
        W = (torch.randn(N, K, generator=g) / 16).to(dev); b = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(dev).to(BF) if hr else None
        w = ops.strip_pack(W)
        outs = {}
        for form in ('0', '1'):
            os.environ[form_var] = form
            y = ops.strip_linear(x, w, N, bias=b, residual=res)
            torch.cuda.synchronize()
            t = []
            for _ in range(3):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): y = ops.strip_linear(x, w, N, bias=b, residual=res)
                e1.record(); torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) / 5 * 1e3)
            outs[form] = (y.clone(), min(t))
        same = torch.equal(outs['0'][0], outs['1'][0])
        ok &= same
        nbad = (outs['0'][0] != outs['1'][0]).sum().item()
        print(f'M={M:7d} N={N} K={K} res={int(hr)}: default {outs["0"][1]:7.1f} us  {form_var[-2:]} {outs["1"][1]:7.1f} us  identical={same} (differing {nbad})', flush=True)
os.environ[form_var] = '0'
print('ALL IDENTICAL' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
