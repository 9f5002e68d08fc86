"""x3 / exact-fp32 ('parity') / bf16 outputs of a trained checkpoint on the first clips of the scored minute, saved for a comparison against
an fp64 evaluation of the oracle graph on the CPU (tools/dbg_trained_parity_cpu.py): which mode is how far from the TRUE value?"""
import os, pickle, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import numpy as np, torch
from corpus import synth_audio as SA
from model.amt import AMT
pkl, out, n_clips = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
amt = AMT(SA.default_config(), pkl, batch_size=8)
notes = SA.pluck_notes(1234)
feat = amt.wave2feature(SA.pluck_wave(notes, device=dev).unsqueeze(0), SA.SR).numpy()      # (the waveform tools/train_config5.py scores)
cin = SA.default_config()['input']
pad = np.full((cin['margin_b'], feat.shape[1]), cin['min_value'], np.float32)
a = np.concatenate([pad, feat, np.full((192, feat.shape[1]), cin['min_value'], np.float32)], 0)
starts = [i * 128 for i in range(min(n_clips, (a.shape[0] - 192) // 128 + 1))]
spec = np.stack([a[s:s + 192].T for s in starts]).astype(np.float32)          # [B, 256, 192]
res = {'spec': spec}
model = amt.model
for mode in ('parity', 'x3', 'bf16'):
    model.hftt_precision = mode
    model.eval()
    with torch.no_grad():
        o = model(torch.from_numpy(spec).to(dev))
    for k, t in enumerate(o):
        if k not in (3, 4, 8):                        # (posteriors only: the two velocity-logit tensors are 17 MB per clip and mode)
            res['%s.%d' % (mode, k)] = t.cpu().numpy()
np.savez_compressed(out, **res)
print('saved', out, {k: v.shape for k, v in res.items() if k.startswith('x3')})
