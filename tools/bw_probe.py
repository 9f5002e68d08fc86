import torch, time
dev = torch.device('cuda:0')
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e-3)
    return best
for mb in (134, 268, 537, 1074):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x)
    xf = x.view(torch.int32)
    s = t(lambda: xf.sum())
    c = t(lambda: y.copy_(x))
    f = t(lambda: y.fill_(1.0))
    a = t(lambda: torch.add(x, x, out=y))
    print(f'{mb} MB: read(sum int32) {mb/1e6/s:.2f} TB/s  copy {2*mb/1e6/c:.2f} TB/s (r+w)  fill {mb/1e6/f:.2f} TB/s  add(x,x) {2*mb/1e6/a:.2f} TB/s', flush=True)
