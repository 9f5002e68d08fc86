"""Shared helpers for the tests (test infrastructure; may import oracle/)."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'nylon-amt_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import hftt_oracle as O   # noqa: E402

MINI = O.HfttConfig(n_margin=4, n_frame=16, n_bin=48, cnn_channel=4, cnn_kernel=5, hid_dim=64, pf_dim=96,
                    enc_layer=2, dec_layer=2, enc_head=2, dec_head=2, n_note=12, n_velocity=16)

OUT_NAMES = ['onset_A', 'offset_A', 'mpe_A', 'velocity_A', 'attention', 'onset_B', 'offset_B', 'mpe_B', 'velocity_B']


def initialize_weights(m):   # call restated from training/m_training.py:31-33
    if hasattr(m, 'weight') and m.weight.dim() > 1:
        nn.init.xavier_uniform_(m.weight.data)


def build_model(cfg, seed, dropout=0.0, device='cpu'):
    """Replays m_training.py:109-141: seed, positional construction, .to(device), apply(initialize_weights)."""
    from model.model_spec2midi import Encoder_SPEC2MIDI, Decoder_SPEC2MIDI, Model_SPEC2MIDI
    torch.manual_seed(seed)
    enc = Encoder_SPEC2MIDI(cfg.n_margin, cfg.n_frame, cfg.n_bin, cfg.cnn_channel, cfg.cnn_kernel, cfg.hid_dim,
                            cfg.enc_layer, cfg.enc_head, cfg.pf_dim, dropout, device)
    dec = Decoder_SPEC2MIDI(cfg.n_frame, cfg.n_bin, cfg.n_note, cfg.n_velocity, cfg.hid_dim, cfg.dec_layer,
                            cfg.dec_head, cfg.pf_dim, dropout, device)
    model = Model_SPEC2MIDI(enc, dec)
    model.apply(initialize_weights)      # on CPU, exactly like make_golden.py (same RNG stream as the reference)
    return model


def perturb(model, seed):
    """Same perturbation as tests/golden/make_golden.py (non-trivial LayerNorm affine and biases)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                if name.endswith('layer_norm.weight'):
                    p.add_(0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def cfg_from_golden(g):
    keys = list(O.PAPER.as_dict().keys())
    return O.HfttConfig(**{k: int(v) for k, v in zip(keys, g['cfg'])})


def sd_cpu(model):
    return {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}


# ---- numpy emulation of the device dropout RNG (csrc/hftt_common.h: hftt_hash / hftt_keep_thr) ----
def keep_mask(seed, site, idx, p):
    """numpy emulation of hftt_hash / hftt_keep (csrc/hftt_common.h): splitmix64 key over (seed, site), one 32-bit mixer per
    QUAD of elements (idx >> 2); element idx takes byte (idx & 3) of the word, compared with round((1-p)*256)."""
    M64, M32 = np.uint64, np.uint32
    with np.errstate(over='ignore'):
        k = M64(seed) + M64(0x9E3779B97F4A7C15) * M64(site + 1)
        k ^= k >> M64(30); k *= M64(0xBF58476D1CE4E5B9)
        k ^= k >> M64(27); k *= M64(0x94D049BB133111EB)
        k ^= k >> M64(31)
        idx = idx.astype(np.uint64)
        q = idx >> M64(2)
        lo = (q & M64(0xFFFFFFFF)).astype(np.uint32); hi = (q >> M64(32)).astype(np.uint32)
        x = (lo + M32(int(k) & 0xFFFFFFFF)) ^ (hi ^ (hi << M32(16)))
        x ^= x >> M32(16); x *= M32(0x7FEB352D)
        x ^= x >> M32(15); x ^= M32(int(k) >> 32); x *= M32(0x846CA68B)
        x ^= x >> M32(16)
    field = ((x >> (M32(8) * (idx & M64(3)).astype(np.uint32))) & M32(0xFF)).astype(np.uint32)
    kk = (1.0 - float(np.float32(p))) * 256.0 + 0.5
    thr = 256 if kk >= 256.0 else (0 if kk <= 0.0 else int(kk))
    return field < np.uint32(thr)


def keep_scale(p):
    """scale of the kept elements (csrc/hftt_common.h: hftt_keep_scale): 256 / thr, the reciprocal of the keep probability actually applied"""
    if not p > 0.0:
        return 1.0
    kk = (1.0 - float(np.float32(p))) * 256.0 + 0.5
    thr = 256 if kk >= 256.0 else (0 if kk <= 0.0 else int(kk))
    return float(np.float32(256.0) / np.float32(thr)) if thr > 0 else 0.0


def keep_mask_t(seed, site, shape, p):
    n = int(np.prod(shape))
    return torch.from_numpy(keep_mask(seed, site, np.arange(n, dtype=np.uint64), p).reshape(shape))


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def max_err(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()
