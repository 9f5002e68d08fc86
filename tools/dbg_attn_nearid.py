import sys, math, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, 'nylon-amt_amd')
from hftt_hip import ops
from util import keep_mask_t, keep_scale
dev = torch.device('cuda:0')
bf = torch.bfloat16
def ref(q, k, v, H, mask, sc):
    n, Lq, d = q.shape; Lk = k.shape[1]; dh = d // H
    qh = q.view(n, Lq, H, dh).transpose(1, 2); kh = k.view(n, Lk, H, dh).transpose(1, 2); vh = v.view(n, Lk, H, dh).transpose(1, 2)
    pr = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), -1)
    pd = pr if mask is None else pr * mask * sc
    return (pd @ vh).transpose(1, 2).reshape(n, Lq, d)
for (n, H, Lq, Lk, dh) in ((64, 4, 16, 16, 64), (16, 4, 128, 128, 64), (16, 2, 48, 48, 32)):
  for spread in (1.0, 0.02):
    for p in (0.0, 0.1):
        d = H * dh
        g = torch.Generator().manual_seed(5)
        q = torch.randn(n, Lq, d, generator=g).to(bf)
        kbar = torch.randn(n, 1, d, generator=g) * 3
        k = (kbar + spread * torch.randn(n, Lk, d, generator=g)).to(bf)
        v = (torch.randn(n, 1, d, generator=g) * 3 + torch.randn(n, Lk, d, generator=g)).to(bf)
        do = torch.randn(n, Lq, d, generator=g).to(bf)
        site, seed = 5, 777
        mask = keep_mask_t(seed, site, (n, H, Lq, Lk), p).double() if p > 0 else None
        q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
        o = ref(q64, k64, v64, H, mask, keep_scale(p)); (o * do.double()).sum().backward()
        res = {}
        for mode in ('bf16', 'x3'):
            if mode == 'bf16':
                a = [t.to(dev) for t in (q, k, v)]
                out, lse = ops.attn_fwd(*a, H, npass=1, drop_p=p, drop_site=site, drop_seed=seed, out_dtype=bf)
                dq, dk, dv = ops.attn_bwd(*a, out, lse, do.to(dev), H, npass=1, drop_p=p, drop_site=site, drop_seed=seed, dq_dtype=bf, dkv_dtype=bf)
            else:
                a = [t.float().to(dev) for t in (q, k, v)]
                out, lse = ops.attn_fwd(*a, H, npass=2, drop_p=p, drop_site=site, drop_seed=seed)
                dq, dk, dv = ops.attn_bwd(*a, out, lse, do.float().to(dev), H, npass=2, drop_p=p, drop_site=site, drop_seed=seed)
            def cos(a_, b_):
                a_ = a_.double().cpu().flatten(); b_ = b_.flatten()
                return float(a_ @ b_ / (a_.norm() * b_.norm()))
            res[mode] = (cos(dq, q64.grad), cos(dk, k64.grad), cos(dv, v64.grad))
        print((n, H, Lq, Lk, dh), 'spread', spread, 'p', p, {m: [round(c, 4) for c in r] for m, r in res.items()}, flush=True)
