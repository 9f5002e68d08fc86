"""evaluation.metrics: hand-computed known answers and properties.  mir_eval is absent and unpinned, so
these pin the restated definitions, not mir_eval's output (the module header says "parity unpinned" at that boundary)."""
import numpy as np
import pytest

from evaluation.metrics import frame_metrics, note_metrics, _max_matching


def N(pitch, onset, offset):
    return {'pitch': pitch, 'onset': onset, 'offset': offset, 'velocity': 64}


def test_frame_metrics_known_answer():
    ref = np.zeros((4, 5), bool); est = np.zeros((5, 5), bool)          # est one frame longer: ignored
    ref[0, [0, 1]] = True; ref[1, 2] = True; ref[3, 4] = True              # 4 reference pitches
    est[0, [1, 3]] = True; est[1, 2] = True; est[2, 0] = True; est[4, :] = True   # 4 estimated in the common frames
    m = frame_metrics(ref, est)
    assert (m['n_ref'], m['n_est'], m['n_correct']) == (4, 4, 2)
    assert m['Precision'] == 0.5 and m['Recall'] == 0.5 and m['f1'] == 0.5
    assert m['Accuracy'] == pytest.approx(2 / (2 + 2 + 2))
    post = np.array([[0.5, 0.49], [0.7, 0.1]])
    assert frame_metrics(post >= 0.5, post, threshold=0.5)['f1'] == 1.0    # `>=` like m_mpe.py:102
    z = frame_metrics(np.zeros((3, 2), bool), np.zeros((3, 2), bool))
    assert z['f1'] == 0.0 and z['Accuracy'] == 0.0                         # empty rolls: zero, not a division error
    with pytest.raises(ValueError):
        frame_metrics(np.zeros((3, 2)), np.zeros((3, 3)))


def test_note_metrics_known_answers():
    ref = [N(60, 1.00, 1.50), N(64, 2.00, 2.40), N(67, 3.00, 4.00)]
    est = [N(60, 1.04, 1.90),            # onset +40 ms: match; offset off by 0.40 > max(0.05, 0.2*0.5) -> fails with offsets
           N(64, 2.06, 2.40),            # onset +60 ms: no match
           N(67, 2.95, 4.15),            # onset -50 ms (edge counts), offset +0.15 <= 0.2*1.0 -> match either way
           N(72, 5.00, 5.00)]            # zero length: dropped before scoring
    m = note_metrics(ref, est)
    assert (m['n_ref'], m['n_est'], m['n_matched']) == (3, 3, 2)
    assert m['Precision'] == pytest.approx(2 / 3) and m['Recall'] == pytest.approx(2 / 3) and m['F-measure'] == pytest.approx(2 / 3)
    mo = note_metrics(ref, est, with_offset=True)
    assert mo['n_matched'] == 1 and mo['F-measure'] == pytest.approx(1 / 3)
    assert note_metrics(ref, ref)['F-measure'] == 1.0 and note_metrics(ref, ref, with_offset=True)['F-measure'] == 1.0
    assert note_metrics(ref, [])['F-measure'] == 0.0 and note_metrics([], est)['F-measure'] == 0.0
    assert note_metrics(ref, [N(61, 1.0, 1.5)])['n_matched'] == 0          # a semitone off is a different note


def test_note_matching_is_maximum_not_greedy():
    # two reference notes 60 ms apart, two estimates: e0 is within tolerance of both, e1 only of r0.  A greedy pass that gives
    # e0 to r0 leaves r1 unmatched (1 match); the maximum matching pairs r0-e1, r1-e0 (2 matches).
    ref = [N(60, 1.00, 1.2), N(60, 1.06, 1.3)]
    est = [N(60, 1.03, 1.2), N(60, 0.96, 1.1)]
    m = note_metrics(ref, est)
    assert m['n_matched'] == 2 and sorted(m['matching']) == [(0, 1), (1, 0)]
    # one estimate can serve one reference only
    assert note_metrics([N(60, 1.0, 1.2), N(60, 1.01, 1.2)], [N(60, 1.0, 1.2)])['n_matched'] == 1
    size, _ = _max_matching([[0, 1], [0], [1, 2]], 3)
    assert size == 3


def test_decode_and_score_round_trip_on_rendered_rolls():
    """Posteriorgrams rendered from a note list -> AMT.mpe2note -> note_metrics against the list: F-measure 1.0.  Pins the decoder
    (section 8(f) #1) and the scorer together without any model."""
    from model.amt import AMT
    cfg = {'feature': {'sr': 16000, 'hop_sample': 256, 'n_bins': 256}, 'input': {'margin_b': 32, 'margin_f': 32, 'num_frame': 128},
           'midi': {'note_min': 21, 'num_note': 88, 'num_velocity': 128}}
    amt = AMT(cfg, None)
    hop = 256 / 16000.0
    rng = np.random.RandomState(1234)
    notes = []
    t = 0.5
    for _ in range(40):
        pitch = int(rng.randint(40, 89)); dur = float(rng.uniform(0.2, 0.8))
        notes.append(N(pitch, round(t / hop) * hop, round((t + dur) / hop) * hop))   # frame-aligned: exact sub-frame refinement
        t += float(rng.uniform(0.15, 0.5))
    n_frame = int((t + 2.0) / hop)
    onset = np.zeros((n_frame, 88), np.float32); offset = np.zeros_like(onset); mpe = np.zeros_like(onset)
    vel = np.zeros((n_frame, 88), np.int64)
    for n in notes:
        k = n['pitch'] - 21
        a, b = int(round(n['onset'] / hop)), int(round(n['offset'] / hop))
        for d, v in ((-1, 0.6), (0, 1.0), (1, 0.6)):                       # triangular targets like the training labels
            onset[a + d, k] = max(onset[a + d, k], v); offset[b + d, k] = max(offset[b + d, k], v)
        mpe[a:b, k] = 1.0
        vel[a - 1:b + 1, k] = 80
    est = amt.mpe2note(a_onset=onset, a_offset=offset, a_mpe=mpe, a_velocity=vel, thred_onset=0.5, thred_offset=0.5, thred_mpe=0.5,
                       mode_velocity='ignore_zero', mode_offset='shorter')
    m = note_metrics(notes, est)
    assert m['n_est'] == len(notes) and m['F-measure'] == 1.0
    assert note_metrics(notes, est, with_offset=True)['F-measure'] == 1.0
    roll = np.zeros((n_frame, 88), bool)
    for n in est:
        roll[int(round(n['onset'] / hop)):int(round(n['offset'] / hop)), n['pitch'] - 21] = True
    assert frame_metrics(mpe >= 0.5, roll)['f1'] == 1.0
