"""CPU side of tools/dbg_trained_parity.py: the oracle graph in fp64 (and fp32) on the saved clips with the checkpoint's parameters; prints,
per output, the maximum distance of every mode from the fp64 values."""
import os, pickle, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import hftt_oracle as O
pkl, npz = sys.argv[1], sys.argv[2]
d = np.load(npz)
with open(pkl, 'rb') as fh:
    model = pickle.load(fh)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
cfg = O.PAPER if sd['encoder_spec2midi.pos_embedding_freq.weight'].shape[1] == 256 else O.TINY
spec = torch.from_numpy(d['spec'])
names = ['onset_A', 'offset_A', 'mpe_A', 'velocity_A', 'attention', 'onset_B', 'offset_B', 'mpe_B', 'velocity_B']
torch.set_num_threads(8)
with torch.no_grad():
    ref64 = O.model_forward({k: v.double() for k, v in sd.items()}, spec.double(), cfg)
    ref32 = O.model_forward(sd, spec, cfg)
rows = {}
for k in (0, 1, 2, 5, 6, 7):
    r64, r32 = ref64[k].numpy(), ref32[k].double().numpy()
    row = {'reference_fp32_vs_fp64': float(np.abs(r32 - r64).max())}
    for mode in ('parity', 'x3', 'bf16'):
        g = d['%s.%d' % (mode, k)].astype(np.float64)
        row[mode + '_vs_fp64'] = float(np.abs(g - r64).max())
        row[mode + '_vs_reference_fp32'] = float(np.abs(g - r32).max())
        row[mode + '_elements_over_1e-3_vs_reference_fp32'] = int((np.abs(g - r32) > 1e-3).sum())
    rows[names[k]] = row
    print('%-9s' % names[k], '  '.join('%s %.2e' % (a, b) for a, b in row.items() if not a.startswith('bf16')))
import json
print(json.dumps(rows))
