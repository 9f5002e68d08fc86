"""Synthetic plucked-string audio with its generating note list: decaying harmonic plucks at MIDI 40-88,
16 kHz mono.  One place for the recipe so that the training set (tools/train_config5.py), the timed inference run
(tools/config5_inference.py) and the tests draw from the same generator -- with different seeds: the scored file is seed 1234, training
files take seeds the scored file never uses.  The reference has no such generator (it trains on MAESTRO, hftt_code/corpus/*): this is the
stand-in corpus of a box without network; what it feeds follows the reference's formats (note dicts as conv_midi2note.py writes them:
pitch / onset / offset / velocity; labels through corpus.conv_note2label)."""
import numpy as np
import torch

SR = 16000


def pluck_notes(seed, dur=60.0):
    """the seed-1234 recipe of round 1 (kept draw for draw: same notes for the same seed)"""
    rng = np.random.RandomState(seed)
    notes, t = [], 0.25
    while t < dur - 2.0:
        notes.append({'pitch': int(rng.randint(40, 89)), 'onset': t, 'offset': t + float(rng.uniform(0.3, 1.2)), 'velocity': int(rng.randint(40, 110))})
        t += float(rng.uniform(0.08, 0.35))
    return notes


def pluck_wave(notes, dur=60.0, sr=SR, device='cpu'):
    """sum over notes of three decaying harmonics (1, 1/2, 1/4), amplitude velocity / 127 * 0.1, envelope exp(-3 (t - onset)) gated to
    [onset, offset + 0.3 s) -- evaluated on each note's own sample range only (identical samples to the whole-array form of round 1)"""
    n = int(sr * dur)
    wave = torch.zeros(n, dtype=torch.float32, device=device)
    for nt in notes:
        i0 = max(0, int(np.floor(nt['onset'] * sr)) - 1)
        i1 = min(n, int(np.ceil((nt['offset'] + 0.3) * sr)) + 1)
        tt = torch.arange(i0, i1, dtype=torch.float32, device=device) / sr
        f0 = 440.0 * 2.0 ** ((nt['pitch'] - 69) / 12.0)
        env = torch.exp(-3.0 * (tt - nt['onset']).clamp(min=0)) * ((tt >= nt['onset']) & (tt < nt['offset'] + 0.3))
        seg = torch.zeros_like(tt)
        for h, a in ((1, 1.0), (2, 0.5), (3, 0.25)):
            if f0 * h < sr / 2:
                seg += (nt['velocity'] / 127.0) * 0.1 * a * torch.sin(2 * np.pi * f0 * h * tt) * env
        wave[i0:i1] += seg
    return wave


def default_config():
    """hftt_code/corpus/config.json with the two keys make_dataset.py adds (min_value, n_bins)"""
    return {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': 256, 'n_bins': 256, 'fft_bins': 2048, 'window_length': 2048,
                        'log_offset': 1e-8, 'window': 'hann', 'pad_mode': 'constant'},
            'input': {'margin_b': 32, 'margin_f': 32, 'num_frame': 128, 'min_value': -18.420681},
            'midi': {'note_min': 21, 'note_max': 108, 'num_note': 88, 'num_velocity': 128}}


def reference_roll(notes, n_frames, sr=SR, hop=256, note_min=21):
    roll = np.zeros((n_frames, 88), bool)
    for nt in notes:
        roll[int(nt['onset'] * sr / hop):int(nt['offset'] * sr / hop), nt['pitch'] - note_min] = True
    return roll
