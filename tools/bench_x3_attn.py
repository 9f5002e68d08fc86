"""Encoder-shaped x3 attention backward / forward timing (HFTT_X3_ATTN_DEBUG switches single mechanisms off: csrc/x3_attn.hip)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
from hftt_hip import ops
dev = torch.device('cuda:0')
n, H, L, dh = int(os.environ.get('NSEQ', 1024)), 4, int(os.environ.get('L', 256)), 64
d = H * dh
g = torch.Generator().manual_seed(1)
qkv = torch.randn(n * L, 3 * d, generator=g).to(dev).view(n, L, 3 * d)
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
do = torch.randn(n, L, d, generator=g).to(dev)
dqkv = torch.empty_like(qkv)
out, lse = ops.attn_fwd(q, k, v, H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3)
def t(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
fw = t(lambda: ops.attn_fwd(q, k, v, H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3))
bw = t(lambda: ops.attn_bwd(q, k, v, out, lse, do, H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3,
                            grads_out=(dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:])))
print('debug %s: fwd %.1f us, bwd %.1f us' % (os.environ.get('HFTT_X3_ATTN_DEBUG', '0'), fw, bw), flush=True)
# the same launches on f16-pair planes (csrc/x3_attn_pl.hip)
pq = ops.to_planes(qkv.contiguous())
qp, kp, vp = pq[..., :d], pq[..., d:2 * d], pq[..., 2 * d:]
fwp = t(lambda: ops.attn_fwd(qp, kp, vp, H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3, planes=True))
bwp = t(lambda: ops.attn_bwd(qp, kp, vp, out, lse, do, H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3, planes=True,
                             grads_out=(dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:])))
print('planes: fwd %.1f us, bwd %.1f us' % (fwp, bwp), flush=True)
if os.environ.get('CROSS', '1') == '1':      # the decoder's cross attention: 88 queries x 256 keys
    Lq = 88
    qc = torch.randn(n, Lq, d, generator=g).to(dev); kv = torch.randn(n, L, 2 * d, generator=g).to(dev)
    doc = torch.randn(n, Lq, d, generator=g).to(dev)
    oc, lc = ops.attn_fwd(qc, kv[..., :d], kv[..., d:], H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3)
    fc = t(lambda: ops.attn_fwd(qc, kv[..., :d], kv[..., d:], H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3))
    qcp = ops.to_planes(qc); kvp = ops.to_planes(kv)
    fcp = t(lambda: ops.attn_fwd(qcp, kvp[..., :d], kvp[..., d:], H, npass=2, drop_p=0.1, drop_site=1, drop_seed=3, planes=True))
    print('cross 88 x %d: fwd %.1f us, planes fwd %.1f us' % (L, fc, fcp), flush=True)
