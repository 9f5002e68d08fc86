#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REFERENCE (build container only).

Run once here:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
It imports /root/reference/hftt_code/model/model_spec2midi.py unmodified, runs it on CPU in fp32 and
writes small .npz fixtures next to this script.  Nothing of the reference's text is stored: fixtures are
inputs, parameters and outputs only.  The GPU box never runs this script (no /root/reference there).

Fixtures
  micro.npz     full tensors: state_dict, input, labels, 9 outputs, loss, every gradient (dropout 0), and the
                parameters after one torch.optim.Adam(lr=1e-4) step.
  tiny_b2.npz   tiny config (m_training.py:55-60 defaults), B=2: seed, per-parameter checksums, strided output
  paper_b1.npz  samples + per-tensor sum/absmax, loss, per-parameter gradient sum/absmax (paper: d256/ff512/3+3/4h).
"""
import os
import sys
import math

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, '/root/reference/hftt_code')
sys.dont_write_bytecode = True

from model.model_spec2midi import Encoder_SPEC2MIDI, Decoder_SPEC2MIDI, Model_SPEC2MIDI   # the REFERENCE  # noqa: E402
assert Model_SPEC2MIDI.__module__ == 'model.model_spec2midi'
import importlib  # noqa: E402
assert importlib.import_module('model.model_spec2midi').__file__.startswith('/root/reference/'), 'must import the reference'

from oracle.hftt_oracle import MICRO, TINY, PAPER, synth_spec, synth_labels   # noqa: E402  (input generators only)


def initialize_weights(m):   # restated call of training/m_training.py:31-33
    if hasattr(m, 'weight') and m.weight.dim() > 1:
        nn.init.xavier_uniform_(m.weight.data)


def build_reference(cfg, seed, dropout=0.0):
    torch.manual_seed(seed)
    enc = Encoder_SPEC2MIDI(cfg.n_margin, cfg.n_frame, cfg.n_bin, cfg.cnn_channel, cfg.cnn_kernel, cfg.hid_dim,
                            cfg.enc_layer, cfg.enc_head, cfg.pf_dim, dropout, 'cpu')
    dec = Decoder_SPEC2MIDI(cfg.n_frame, cfg.n_bin, cfg.n_note, cfg.n_velocity, cfg.hid_dim, cfg.dec_layer,
                            cfg.dec_head, cfg.pf_dim, dropout, 'cpu')
    model = Model_SPEC2MIDI(enc, dec)
    model.apply(initialize_weights)
    return model


def perturb(model, seed):
    """Give LayerNorm affine and all biases non-trivial values so the fixtures exercise them (xavier leaves
    gamma=1, beta=0, and default bias inits are small)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                if name.endswith('layer_norm.weight'):
                    p.add_(0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))


def reference_loss(out, labels, wA=1.0, wB=1.0):
    """training/train.py:106-153 with nn.BCELoss / nn.CrossEntropyLoss exactly as m_training.py:149-157."""
    on_a, of_a, mp_a, ve_a, _att, on_b, of_b, mp_b, ve_b = out
    lo, lf, lm, lv = (t.contiguous().view(-1) for t in labels)
    bce, ce = nn.BCELoss(), nn.CrossEntropyLoss()
    la = bce(on_a.contiguous().view(-1), lo) + bce(of_a.contiguous().view(-1), lf) + bce(mp_a.contiguous().view(-1), lm) \
        + ce(ve_a.contiguous().view(-1, ve_a.shape[-1]), lv)
    lb = bce(on_b.contiguous().view(-1), lo) + bce(of_b.contiguous().view(-1), lf) + bce(mp_b.contiguous().view(-1), lm) \
        + ce(ve_b.contiguous().view(-1, ve_b.shape[-1]), lv)
    return wA * la + wB * lb


OUT_NAMES = ['onset_A', 'offset_A', 'mpe_A', 'velocity_A', 'attention', 'onset_B', 'offset_B', 'mpe_B', 'velocity_B']


def sample_stride(numel):
    return max(1, numel // 4096) | 1


def make_micro():
    cfg = MICRO
    model = build_reference(cfg, seed=1234)
    perturb(model, 99)
    x = synth_spec(2, cfg, salt=11) * 0.25          # keep activations moderate for the 16-wide model
    labels = synth_labels(2, cfg, salt=12)
    model.train()                                    # dropout = 0.0 -> identical to eval
    out = model(x)
    loss = reference_loss(out, labels, 1.0, 0.7)
    loss.backward()
    d = {'cfg': np.array(list(cfg.as_dict().values()), dtype=np.int64), 'input': x.numpy(), 'loss': np.float64(loss.item()),
         'weight_A': np.float64(1.0), 'weight_B': np.float64(0.7)}
    for n, t in zip(['label_onset', 'label_offset', 'label_mpe', 'label_velocity'], labels):
        d[n] = t.numpy()
    for n, t in zip(OUT_NAMES, out):
        d['out.' + n] = t.detach().numpy()
    for k, v in model.state_dict().items():
        d['sd.' + k] = v.detach().numpy().copy()
    for k, p in model.named_parameters():
        d['grad.' + k] = p.grad.detach().numpy().copy()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    opt.step()
    for k, v in model.state_dict().items():
        d['adam1.' + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'micro.npz'), **d)
    print('micro: loss', loss.item(), 'keys', len(d))


def make_big(name, cfg, bsz, seed, with_grads=True):
    model = build_reference(cfg, seed=seed)
    perturb(model, seed + 1)
    x = synth_spec(bsz, cfg, salt=seed)
    labels = synth_labels(bsz, cfg, salt=seed + 7)
    model.train()
    d = {'cfg': np.array(list(cfg.as_dict().values()), dtype=np.int64), 'bsz': np.int64(bsz), 'seed': np.int64(seed)}
    for k, v in model.state_dict().items():
        v64 = v.detach().double()
        d['sdsum.' + k] = np.array([v64.sum().item(), v64.abs().sum().item()])
    if with_grads:
        out = model(x)
        loss = reference_loss(out, labels)
        loss.backward()
    else:
        with torch.no_grad():
            out = model(x)
            loss = reference_loss(out, labels)
    d['loss'] = np.float64(loss.item())
    for n, t in zip(OUT_NAMES, out):
        t = t.detach().reshape(-1)
        st = sample_stride(t.numel())
        d['out.' + n + '.stride'] = np.int64(st)
        d['out.' + n + '.sample'] = t[::st].numpy().copy()
        d['out.' + n + '.stats'] = np.array([t.double().sum().item(), t.abs().max().item()])
    if with_grads:
        for k, p in model.named_parameters():
            g = p.grad.detach().reshape(-1)
            st = max(1, g.numel() // 64) | 1
            d['grad.' + k + '.sample'] = g[::st].numpy().copy()
            d['grad.' + k + '.stats'] = np.array([g.double().sum().item(), g.abs().max().item(), g.double().norm().item()])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)
    print(name, ': loss', loss.item(), 'keys', len(d))


class _EchoModel:
    """Stand-in 'model' for the windowing goldens: a deterministic function of the clip's centre frames."""
    def __init__(self, cfg):
        self.cfg = cfg

    def eval(self):
        return self

    def __call__(self, spec):
        c = self.cfg
        ctr = spec[:, :c['midi']['num_note'], c['input']['margin_b']:c['input']['margin_b'] + c['input']['num_frame']].transpose(1, 2)
        edge = spec[:, :c['midi']['num_note'], :c['input']['num_frame']].transpose(1, 2)     # looks into the left margin
        nv = c['midi']['num_velocity']
        vel = torch.stack([(ctr * (k + 1)).sin() for k in range(nv)], dim=-1)
        return (ctr, ctr * 0.5 + edge, ctr - 1.0, vel, None, edge, ctr + edge, ctr * 2.0, vel.flip(-1))


def make_amt():
    """AMT.transcript / transcript_stride / mpe2note goldens.  model/amt.py imports torchaudio and pretty_midi at module
    top (amt.py:6-7); neither is installed here and neither is touched by the three methods exercised, so EMPTY
    placeholder modules satisfy the import statements (they implement nothing)."""
    import types
    for name in ('torchaudio', 'pretty_midi'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from model.amt import AMT
    assert importlib.import_module('model.amt').__file__.startswith('/root/reference/')
    cfg = {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': 12, 'n_bins': 12}, 'input': {'margin_b': 2, 'margin_f': 2, 'num_frame': 8, 'min_value': -18.5},
           'midi': {'note_min': 21, 'note_max': 28, 'num_note': 8, 'num_velocity': 4}}
    amt = AMT(cfg, None)
    amt.model = _EchoModel(cfg)
    amt.device = 'cpu'
    d = {}
    rng = np.random.RandomState(7)
    for n in (8, 21, 30):
        feat = rng.randn(n, 12).astype(np.float32)
        d[f'tr.{n}.feature'] = feat
        for i, o in enumerate(amt.transcript(feat)):
            d[f'tr.{n}.out{i}'] = o
        for n_off in (0, 2, 4):
            for i, o in enumerate(amt.transcript_stride(feat, n_off)):
                d[f'trs.{n}.{n_off}.out{i}'] = o
    # mpe2note: smooth random posteriorgrams with plateaus and ties, full-size config constants
    cfg2 = {'feature': {'sr': 16000, 'hop_sample': 256}, 'midi': {'note_min': 21, 'num_note': 88}}
    amt2 = AMT(cfg2, None)
    for case in range(2):
        r = np.random.RandomState(100 + case)
        n = 160
        def track():
            x = r.rand(n + 8, 88).astype(np.float32)
            k = np.ones(5, np.float32) / 5
            x = np.stack([np.convolve(x[:, j], k, mode='valid') for j in range(88)], 1)[:n]
            x = (x - x.min()) / (x.max() - x.min())
            x = np.round(x * 20) / 20            # quantise -> plateaus and exact ties
            return x.astype(np.float32)
        on, off, mpe = track(), track(), track()
        vel = r.randint(0, 128, size=(n, 88)).astype(np.int8)
        vel[r.rand(n, 88) < 0.1] = 0
        d[f'm2n.{case}.onset'], d[f'm2n.{case}.offset'], d[f'm2n.{case}.mpe'], d[f'm2n.{case}.velocity'] = on, off, mpe, vel
        for mv in ('ignore_zero', 'org'):
            for mo in ('shorter', 'longer', 'offset'):
                notes = amt2.mpe2note(a_onset=on, a_offset=off, a_mpe=mpe, a_velocity=vel, thred_onset=0.6, thred_offset=0.55, thred_mpe=0.5,
                                      mode_velocity=mv, mode_offset=mo)
                arr = np.array([[x['pitch'], x['onset'], x['offset'], x['velocity']] for x in notes], dtype=np.float64).reshape(-1, 4)
                d[f'm2n.{case}.{mv}.{mo}'] = arr
    np.savez_compressed(os.path.join(HERE, 'amt.npz'), **d)
    print('amt: keys', len(d), 'notes in case0:', len(d['m2n.0.ignore_zero.shorter']))


if __name__ == '__main__':
    torch.set_num_threads(8)
    if '--amt-only' in sys.argv:
        make_amt()
        sys.exit(0)
    make_amt()
    make_micro()
    make_big('tiny_b2', TINY, 2, 4321)
    make_big('paper_b1', PAPER, 1, 2468)
