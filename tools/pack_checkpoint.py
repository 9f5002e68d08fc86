#!/usr/bin/env python3
"""Pack a pickled model (m_training.py:372-373 format) into a small fixture: every matrix as int8 with one fp32 scale per output row,
every vector / position table as fp16 -- np.savez_compressed.  The fixture DEFINES its model: `unpack_state_dict` returns exactly
representable fp32 tensors, and every consumer (the GPU test, the oracle run that made the golden outputs) loads those, so the rounding of
the packing is part of the checkpoint, not an error of any path.  5.5 M parameters -> about 5 MB.

  python tools/pack_checkpoint.py gpurun_out/config5_paper.pkl tests/golden/config5_paper_trained.npz"""
import os, pickle, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import numpy as np
import torch


def pack_state_dict(sd):
    out = {}
    for k, v in sd.items():
        a = v.detach().cpu().float().numpy()
        if a.ndim >= 2 and 'pos_embedding' not in k and a.size >= 4096:
            m = a.reshape(a.shape[0], -1)
            scale = np.abs(m).max(axis=1, keepdims=True) / 127.0
            scale[scale == 0] = 1.0
            out['q8:' + k] = np.clip(np.rint(m / scale), -127, 127).astype(np.int8).reshape(a.shape)
            out['sc:' + k] = scale.astype(np.float32).reshape(-1)
        else:
            out['h16:' + k] = a.astype(np.float16)
    return out


def unpack_state_dict(npz):
    sd = {}
    for key in npz.files:
        kind, k = key.split(':', 1)
        if kind == 'q8':
            q = npz[key].astype(np.float32)
            sd[k] = torch.from_numpy(q * npz['sc:' + k].reshape((-1,) + (1,) * (q.ndim - 1)))
        elif kind == 'h16':
            sd[k] = torch.from_numpy(npz[key].astype(np.float32))
    return sd


def main():
    src, dst = sys.argv[1], sys.argv[2]
    with open(src, 'rb') as fh:
        model = pickle.load(fh)
    sd = model.state_dict()
    packed = pack_state_dict(sd)
    np.savez_compressed(dst, **packed)
    back = unpack_state_dict(np.load(dst))
    worst = max(float((back[k] - v.float()).abs().max() / (v.float().abs().max() + 1e-30)) for k, v in sd.items())
    print('%s: %d tensors, %d parameters, %.2f MB -> %s %.2f MB; worst |delta| / max|w| per tensor %.4f' %
          (src, len(sd), sum(v.numel() for v in sd.values()), os.path.getsize(src) / 1e6, dst, os.path.getsize(dst) / 1e6, worst))


if __name__ == '__main__':
    main()
