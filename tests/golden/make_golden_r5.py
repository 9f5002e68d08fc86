#!/usr/bin/env python3
"""Round-5 golden vectors (build container only; imports THE REFERENCE): config 5 at PAPER size on TRAINED weights.

  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r5.py

Input: tests/golden/config5_paper_trained.npz -- the paper-size model (d 256, ff 512, 3+3 layers, 4 heads) trained by this repo's own
training step on the synthetic plucked-string corpus (recipe: profiles/r05_config5_paper_trained.json `command`), packed by
tools/pack_checkpoint.py (int8 matrices + per-row scales, fp16 vectors: the unpacked tensors ARE the checkpoint).
What it writes (config5_paper_golden.npz): for two of the 30 clips of the scored minute (seed 1234; log-mel by the CPU oracle's restatement,
windows exactly as model/amt.py:66-118 cuts them) the input windows, the reference module's six posterior tensors in full, strided samples +
statistics of the two velocity-logit tensors and of the attention tensor, and the velocity argmax.  The CPU oracle is checked against the
reference on the same weights on the way (it must agree to 2e-6).  Nothing of the reference's text is stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, '..', '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, '/root/reference/hftt_code')
sys.dont_write_bytecode = True

from model.model_spec2midi import Encoder_SPEC2MIDI, Decoder_SPEC2MIDI, Model_SPEC2MIDI   # the REFERENCE  # noqa: E402
import importlib  # noqa: E402
assert importlib.import_module('model.model_spec2midi').__file__.startswith('/root/reference/'), 'must import the reference'

import importlib.util  # noqa: E402
from oracle import hftt_oracle as O   # noqa: E402
from pack_checkpoint import unpack_state_dict   # noqa: E402

# the audio generator is this repo's (the reference has none): load it by path, the package name `corpus` also exists in the reference tree
_spec = importlib.util.spec_from_file_location('hftt_synth_audio', os.path.join(ROOT, 'nylon-amt_amd', 'corpus', 'synth_audio.py'))
SA = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(SA)

CLIPS = (5, 17)
OUT_NAMES = ['onset_A', 'offset_A', 'mpe_A', 'velocity_A', 'attention', 'onset_B', 'offset_B', 'mpe_B', 'velocity_B']


def main():
    torch.set_num_threads(8)
    cfg = O.PAPER
    sd = unpack_state_dict(np.load(os.path.join(HERE, 'config5_paper_trained.npz')))
    enc = Encoder_SPEC2MIDI(cfg.n_margin, cfg.n_frame, cfg.n_bin, cfg.cnn_channel, cfg.cnn_kernel, cfg.hid_dim, cfg.enc_layer, cfg.enc_head,
                            cfg.pf_dim, 0.1, 'cpu')
    dec = Decoder_SPEC2MIDI(cfg.n_frame, cfg.n_bin, cfg.n_note, cfg.n_velocity, cfg.hid_dim, cfg.dec_layer, cfg.dec_head, cfg.pf_dim, 0.1, 'cpu')
    model = Model_SPEC2MIDI(enc, dec)
    model.load_state_dict(sd)
    model.eval()
    notes = SA.pluck_notes(1234)
    feat = O.logmel(SA.pluck_wave(notes))                                  # [3751, 256]
    assert tuple(feat.shape) == (3751, 256), feat.shape
    # model/amt.py:66-118: margin_b frames of min_value in front, windows of margin_b + num_frame + margin_f every num_frame frames
    mv = -18.420681
    T, M = cfg.n_frame, cfg.n_margin
    n_clip = -(-feat.shape[0] // T)
    pad = torch.full((M + n_clip * T + M, feat.shape[1]), mv)
    pad[M:M + feat.shape[0]] = feat
    x = torch.stack([pad[k * T:k * T + T + 2 * M].T for k in CLIPS]).contiguous()      # [2, 256, 192]
    with torch.no_grad():
        out = model(x)
        oo = O.model_forward(sd, x, cfg)
    worst = max(float((a - b).abs().max()) for a, b in zip(out, oo))
    assert worst < 2e-6 * max(1.0, float(out[3].abs().max())), worst
    d = {'clips': np.array(CLIPS), 'input': x.numpy(), 'oracle_vs_reference_max_abs': np.float64(worst)}
    for n, t in zip(OUT_NAMES, out):
        t = t.detach()
        if n in ('velocity_A', 'velocity_B', 'attention'):
            f = t.reshape(-1)
            st = max(1, f.numel() // 16384) | 1
            d['out.' + n + '.stride'] = np.int64(st)
            d['out.' + n + '.sample'] = f[::st].numpy().copy()
            d['out.' + n + '.stats'] = np.array([f.double().sum().item(), f.abs().max().item()])
            if n != 'attention':
                d['out.' + n + '.argmax'] = t.argmax(-1).to(torch.int16).numpy()
                top2 = t.topk(2, dim=-1).values
                d['out.' + n + '.margin'] = (top2[..., 0] - top2[..., 1]).numpy().astype(np.float32)     # how decided the argmax is
        else:
            d['out.' + n] = t.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'config5_paper_golden.npz'), **d)
    on = float((out[7] >= 0.5).float().mean())
    print('config5_paper_golden.npz: clips', CLIPS, 'oracle vs reference', worst, 'active frame fraction (mpe_B)', on,
          'bytes', os.path.getsize(os.path.join(HERE, 'config5_paper_golden.npz')))


if __name__ == '__main__':
    main()
