"""the convergence test's WIDE run (tests/test_convergence_gpu.py) in one precision mode, for A/B runs under environment switches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import torch
import test_convergence_gpu as T
dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'x3'
drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
data = T.make_clips(T.WIDE, 64, seed=1)
held = T.make_clips(T.WIDE, 16, seed=2)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2025
import util
_bm = util.build_model
util.build_model = lambda cfg, s_, dropout=0.0: _bm(cfg, seed, dropout=dropout)      # (the test builds its model from seed 2025)
curve, f1b, f1a, _, auc = T.train_device(T.WIDE, mode, drop, data, held, dev, 4, 3e-4)
print('RESULT', mode, 'seed', seed, 'planes=%s' % os.environ.get('HFTT_X3_PLANES', '1'), [round(c, 4) for c in curve], 'f1_B %.4f f1_A %.4f' % (f1b, f1a))
