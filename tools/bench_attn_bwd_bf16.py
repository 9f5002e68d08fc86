"""bf16-mode attention backward (csrc/attn_bwd.hip, every tensor stored as bf16) at a given (sequences, queries, keys): NSEQ / LQ / LK."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
from hftt_hip import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
n, H, dh = int(os.environ.get('NSEQ', 1024)), 4, 64
Lq, Lk = int(os.environ.get('LQ', 256)), int(os.environ.get('LK', 256))
d = H * dh
g = torch.Generator().manual_seed(1)
q = torch.randn(n, Lq, d, generator=g).to(dev).to(BF); kv = torch.randn(n, Lk, 2 * d, generator=g).to(dev).to(BF); do = torch.randn(n, Lq, d, generator=g).to(dev).to(BF)
k, v = kv[..., :d], kv[..., d:]
kw = dict(npass=1, drop_p=0.1, drop_site=1, drop_seed=3)
out, lse = ops.attn_fwd(q, k, v, H, out_dtype=BF, **kw)
dq = torch.empty_like(q); dkv = torch.empty_like(kv)
def t(fn, reps=8):
    for _ in range(int(os.environ.get('WARM', 2))): fn()      # (WARM=300: steady-state clocks -- the first kernels of a process are clocked lower)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
bw = t(lambda: ops.attn_bwd(q, k, v, out, lse, do, H, grads_out=(dq, dkv[..., :d], dkv[..., d:]), **kw))
print('bf16  nseq %d Lq %d Lk %d: bwd %.1f us' % (n, Lq, Lk, bw), flush=True)
