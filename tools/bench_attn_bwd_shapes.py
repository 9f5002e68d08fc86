"""x3 attention backward on f16-pair planes at a given (sequences, queries, keys): NSEQ / LQ / LK.  Used for the key-halves proxy of DESIGN
section 5 (two 4-wave workgroups per CU on half the keys each = the 128-key kernel on twice the sequences, minus the dQ sum)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
from hftt_hip import ops
dev = torch.device('cuda:0')
n, H, dh = int(os.environ.get('NSEQ', 1024)), 4, 64
Lq, Lk = int(os.environ.get('LQ', 256)), int(os.environ.get('LK', 256))
d = H * dh
g = torch.Generator().manual_seed(1)
q = torch.randn(n, Lq, d, generator=g).to(dev)
kv = torch.randn(n, Lk, 2 * d, generator=g).to(dev)
do = torch.randn(n, Lq, d, generator=g).to(dev)
qp, kvp = ops.to_planes(q), ops.to_planes(kv)
kp, vp = kvp[..., :d], kvp[..., d:]
out, lse = ops.attn_fwd(q, kv[..., :d], kv[..., d:], H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3)      # (fp32 operands: any Lq / Lk; bit-identical outputs)
dq = torch.empty_like(q); dkv = torch.empty_like(kv)
def t(fn, reps=8):
    for _ in range(int(os.environ.get('WARM', 2))): fn()      # (WARM=300: steady-state clocks -- the first kernels of a process are clocked lower)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
bw = t(lambda: ops.attn_bwd(qp, kp, vp, out, lse, do, H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3, planes=True,
                            grads_out=(dq, dkv[..., :d], dkv[..., d:])))
pairs = n * H * ((Lq + 31) // 32) * ((Lk + 31) // 32)
print('debug %s  nseq %d Lq %d Lk %d: bwd %.1f us, %.0f (32 x 32) tile pairs / us' % (os.environ.get('HFTT_X3_ATTN_DEBUG', '0'), n, Lq, Lk, bw, pairs / bw), flush=True)
