"""Dataset assembly and synthetic MAESTRO-format stores.

``assemble_store`` restates the array layout of the reference's ``corpus/make_dataset.py:11-239`` (in memory instead of through per-file
pickles): the corpus is ONE array per tensor, each file preceded / followed by padding so that every clip window of a file stays inside
its own padding -- ``margin_b`` rows first, then per file ``n_i`` rows of data followed by ``margin_f + num_frame - 1`` rows of padding;
``idx`` lists the first frame of every clip (every frame of every file).  The feature padding value is ``log(log_offset)`` (:105-113),
label padding is zero.  ``synth_store`` fills such a store with the synthetic recipe of SURVEY.md section 8(d) config 1."""
import numpy as np


def assemble_store(features, labels, config):
    """features: list of [n_i, mel_bins] fp32; labels: list of dicts with 'onset', 'offset' (fp32 [n_i, notes]), 'mpe' (bool), 'velocity'
    (int8).  -> dict(feature, label_onset, label_offset, label_mpe, label_velocity, idx)."""
    cin, cf, cm = config['input'], config['feature'], config['midi']
    gap = cin['margin_f'] + cin['num_frame'] - 1
    nf = [max(f.shape[0], len(l['mpe'])) for f, l in zip(features, labels)]         # make_dataset.py:52
    total = cin['margin_b'] + sum(n + gap for n in nf)
    zero_value = np.log(cf['log_offset']) if cf['log_offset'] > 0.0 else cf['log_offset']
    # make_dataset.py:100-116: with input.max_value > 0 the features are scaled to (x - min) / (max - min) and the padding is ZERO;
    # otherwise (the shipped config: max_value 0) raw features and log(log_offset) padding
    scaled = cin.get('max_value', 0.0) > 0.0
    if scaled:
        feature = np.zeros([total, cf['mel_bins']], dtype=np.float32)
    else:
        feature = np.full([total, cf['mel_bins']], zero_value, dtype=np.float32)
    lab = {'onset': np.zeros([total, cm['num_note']], np.float32), 'offset': np.zeros([total, cm['num_note']], np.float32),
           'mpe': np.zeros([total, cm['num_note']], bool), 'velocity': np.zeros([total, cm['num_note']], np.int8)}
    idx = np.zeros(sum(nf), dtype=np.int32)
    loc_i, loc_d = 0, cin['margin_b']
    for f, l, n in zip(features, labels, nf):
        idx[loc_i:loc_i + n] = np.arange(loc_d, loc_d + n)
        feature[loc_d:loc_d + f.shape[0]] = ((f - cin['min_value']) / (cin['max_value'] - cin['min_value'])) if scaled else f
        for k in lab:
            lab[k][loc_d:loc_d + len(l[k])] = l[k]
        loc_i += n
        loc_d += n + gap
    return {'feature': feature, 'label_onset': lab['onset'], 'label_offset': lab['offset'], 'label_mpe': lab['mpe'],
            'label_velocity': lab['velocity'], 'idx': idx}


def prepare_config(config, max_value=0.0):
    """What the reference's make_dataset.py __main__ sets BEFORE assembling (:274-278): input.max_value from the command line and
    input.min_value = log(log_offset) as a float32 (or log_offset itself when that is not positive).  In place; returns config."""
    config['input']['max_value'] = max_value
    lo = config['feature']['log_offset']
    config['input']['min_value'] = np.log(lo).astype(np.float32) if lo > 0.0 else lo
    return config


def finalize_config(config):
    """... and what it writes into the config file AFTER assembling (:305-306): min_value as a plain float, feature.n_bins = mel_bins --
    the two keys model/amt.py:70-73 reads when it pads a feature array.  In place; returns config."""
    config['input']['min_value'] = float(config['input']['min_value'])
    config['feature']['n_bins'] = config['feature']['mel_bins']
    return config


def synth_file(n_frames, config, rng):
    """one synthetic 'file': log-mel-like features N(-7, 3^2) clipped to [log(offset), 6]; random notes rendered as the reference's
    label tracks (triangular onset / offset ramps of half-width 3 frames, binary mpe, velocity held over the note)."""
    cf, cm = config['feature'], config['midi']
    lo = float(np.log(cf['log_offset'])) if cf['log_offset'] > 0 else -18.420681
    feat = np.clip(rng.randn(n_frames, cf['mel_bins']).astype(np.float32) * 3.0 - 7.0, lo, 6.0).astype(np.float32)
    N = cm['num_note']
    onset = np.zeros((n_frames, N), np.float32); offset = np.zeros((n_frames, N), np.float32)
    mpe = np.zeros((n_frames, N), bool); vel = np.zeros((n_frames, N), np.int8)
    for _ in range(max(1, n_frames // 12)):
        p, a = int(rng.randint(0, N)), int(rng.randint(0, max(1, n_frames - 4)))
        b = min(n_frames - 1, a + int(rng.randint(2, 40)))
        v = int(rng.randint(1, min(128, cm['num_velocity'])))
        mpe[a:b + 1, p] = True
        vel[a:b + 1, p] = v
        for c, tr in ((a, onset), (b, offset)):
            for k in range(-3, 4):
                if 0 <= c + k < n_frames:
                    tr[c + k, p] = max(tr[c + k, p], 1.0 - abs(k) / 4.0)
    return feat, {'onset': onset, 'offset': offset, 'mpe': mpe, 'velocity': vel}


def synth_store(config, frames_per_file, seed=1234):
    rng = np.random.RandomState(seed)
    files = [synth_file(n, config, rng) for n in frames_per_file]
    return assemble_store([f for f, _ in files], [l for _, l in files], config)
