"""Diagnostic (not a test): how noisy are the parity-mode gradients compared with a CPU fp32 evaluation of the same graph, both measured
against an fp64 evaluation?  Uses the oracle (test infrastructure).  usage: python tests/diag_first_layer_grad.py [tiny_b2|paper_b1]"""
import sys, json
import torch
import util
from util import O

name = sys.argv[1] if len(sys.argv) > 1 else 'tiny_b2'
g = util.golden(name)
cfg = util.cfg_from_golden(g)
seed, B = int(g['seed']), int(g['bsz'])
model = util.build_model(cfg, seed)
util.perturb(model, seed + 1)
x = O.synth_spec(B, cfg, salt=seed)
labels = O.synth_labels(B, cfg, salt=seed + 7)
sd = util.sd_cpu(model)


def oracle_grads(dt):
    p = {k: v.to(dt).clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    outs = O.model_forward(p, x.to(dt), cfg)
    loss = O.spec2midi_loss(outs, *labels)
    loss.backward()
    return {k: v.grad.double() for k, v in p.items() if v.grad is not None}, float(loss)


g64, l64 = oracle_grads(torch.float64)
g32, l32 = oracle_grads(torch.float32)
print('loss fp64 %.9f fp32 %.9f' % (l64, l32))

dev = torch.device('cuda:0')
from hftt_hip.trainer import TrainStep
model = model.to(dev)
model.hftt_precision = 'parity'
model.train()
ts = TrainStep(model)
loss = ts.forward_backward(x.to(dev), *[t.to(dev) for t in labels])
print('loss device %.9f' % loss[0].item())
rows = []
for (pname, _, o, n) in ts.engine._bound:
    gd = ts.engine.flat_grads[o:o + n].cpu().double()
    if pname not in g64:
        continue
    ref = g64[pname].reshape(-1)
    sc = ref.abs().max().item()
    if sc < 1e-12:
        continue
    e_dev = (gd - ref).abs().max().item() / sc
    e_32 = (g32[pname].reshape(-1) - ref).abs().max().item() / sc
    rows.append((e_dev / max(e_32, 1e-12), pname, e_dev, e_32))
rows.sort(reverse=True)
print('%-70s %10s %10s %8s' % ('parameter', 'device', 'cpu fp32', 'ratio'))
for r, pname, e_dev, e_32 in rows[:25]:
    print('%-70s %10.2e %10.2e %8.1f' % (pname, e_dev, e_32, r))
print('median ratio %.2f' % sorted(r[0] for r in rows)[len(rows) // 2])
