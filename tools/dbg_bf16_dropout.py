"""bf16 mode vs parity mode gradients with dropout on (same device masks): per-parameter cosine, to find a dropout site whose forward and
backward masks disagree in one mode."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
import util
from util import O, MINI
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nylon-amt_amd'))
from hftt_hip.trainer import TrainStep

WIDE = O.HfttConfig(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, hid_dim=256, pf_dim=512, enc_layer=1, dec_layer=1,
                    enc_head=4, dec_head=4, n_note=8, n_velocity=16)
dev = torch.device('cuda:0')
for nm, cfg in (('mini', MINI), ('wide', WIDE)):
    for p in (0.0, 0.1):
        B = 4
        x = O.synth_spec(B, cfg, salt=31).to(dev)
        labels = [t.to(dev) for t in O.synth_labels(B, cfg, salt=32)]
        res = {}
        for mode in ('parity', 'bf16', 'x3'):
            model = util.build_model(cfg, 2024, dropout=p)
            if os.environ.get("PERTURB", "1") == "1": util.perturb(model, 2025)
            model = model.to(dev); model.train(); model.hftt_precision = mode
            ts = TrainStep(model)
            loss = ts.forward_backward(x, *labels)
            torch.cuda.synchronize()
            eng = ts.engine
            res[mode] = (loss[0].item(), {name: eng.flat_grads[o:o + n].clone() for (name, _, o, n) in eng._bound})
        for mode in ('bf16', 'x3'):
            cs = []
            for name, gp in res['parity'][1].items():
                if gp.abs().max().item() < 1e-9: continue
                g = res[mode][1][name]
                c = float((g.double() @ gp.double()) / (g.double().norm() * gp.double().norm() + 1e-300))
                cs.append((c, name, float(g.norm() / (gp.norm() + 1e-30))))
            cs.sort()
            print(nm, 'dropout', p, mode, 'loss', res[mode][0], 'parity', res['parity'][0], 'worst:', [(round(c, 4), n, round(r, 3)) for c, n, r in cs[:6]], 'median', cs[len(cs) // 2][0], flush=True)
        if nm == os.environ.get('DETAIL', 'wide'):
            print('   ---- dropout', p)
            for name, gp in res['parity'][1].items():
                g = res['bf16'][1][name]
                c = float((g.double() @ gp.double()) / (g.double().norm() * gp.double().norm() + 1e-300))
                print('   %-70s cos %8.4f  |g| %.3e' % (name, c, float(gp.norm())))
