import os, sys, json
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import numpy as np, torch
from corpus import synth_audio as SA
from model.amt import AMT
dev = torch.device('cuda:0')
pkl = sys.argv[1]
notes = SA.pluck_notes(1234)
wave = SA.pluck_wave(notes, device=dev)
res = {}
for mode in ('parity', 'x3'):
    amt = AMT(SA.default_config(), pkl, batch_size=32)
    amt.model.hftt_precision = mode
    feat = amt.wave2feature(wave.unsqueeze(0), SA.SR)
    res[mode] = amt.transcript(feat.numpy())
    res[mode + '_feat'] = feat.numpy()
print('features equal:', np.array_equal(res['parity_feat'], res['x3_feat']))
names = ['onset_A', 'offset_A', 'mpe_A', 'velocity_A', 'onset_B', 'offset_B', 'mpe_B', 'velocity_B']
for k in (0, 1, 2, 4, 5, 6):
    a, b = res['x3'][k], res['parity'][k]
    d = np.abs(a - b)
    i = np.unravel_index(d.argmax(), d.shape)
    big = np.argwhere(d > 1e-3)
    print(names[k], 'max', d.max(), 'at frame %d note %d' % i, 'x3 %.6f parity %.6f' % (a[i], b[i]), 'elements > 1e-3:', len(big), 'frames:', sorted(set(big[:, 0].tolist()))[:12])
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'c5_feat.npz'), feat=res['x3_feat'])
