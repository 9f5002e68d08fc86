import sys, math, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, 'nylon-amt_amd')
from hftt_hip import ops
dev = torch.device('cuda:0')
n, H, Lq, Lk, dh = 4, 2, 48, 48, 32
g = torch.Generator().manual_seed(11)
d = H * dh
q = torch.randint(-8, 9, (n, Lq, d), generator=g).float() * 2048.0
k = torch.randint(-8, 9, (n, Lk, d), generator=g).float() * 2048.0
v = torch.randn(n, Lk, d, generator=g).bfloat16().float()
e = (q.double().view(n, Lq, H, dh).transpose(1, 2) @ k.double().view(n, Lk, H, dh).transpose(1, 2).transpose(-1, -2))
pr = torch.softmax(e / math.sqrt(dh), -1)
for npass, dt in ((1, torch.bfloat16), (1, torch.float32), (2, torch.float32), (3, torch.float32)):
    out, lse, probs = ops.attn_fwd(q.to(dev).to(dt), k.to(dev).to(dt), v.to(dev).to(dt), H, npass=npass, want_probs=True)
    diff = (probs.cpu().double() - pr).abs()
    i = int(diff.argmax()); idx = list(torch.unravel_index(torch.tensor(i), diff.shape)); idx = [int(t) for t in idx]
    s_, h_, r_, c_ = idx
    row = e[s_, h_, r_]
    top = torch.topk(row, 3)
    print(npass, dt, 'maxdiff', float(diff.max()), 'at', idx, 'ref row top3', top.values.tolist(), top.indices.tolist(),
          'dev probs at those', probs[s_, h_, r_].cpu()[top.indices].tolist(), 'ref', pr[s_, h_, r_][top.indices].tolist(), 'lse', lse[s_, h_, r_].tolist())
