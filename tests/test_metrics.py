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


def test_reshape_for_mir_eval_known_answer():
    """training/train.py:9-57 on a hand-made 2-clip matrix (hop 4410 / 44100 -> 0.1 s frames, so nothing is stretched)."""
    from evaluation.metrics import reshape_for_mir_eval
    on = np.zeros((2, 6, 3)); off = np.zeros((2, 6, 3))
    on[0, [1, 4], 1] = (0.3, 1.0); off[0, [2, 3], 1] = (1.0, 0.2)       # onset 1 -> offset 2; onset 4 -> none after it -> 4 + 1
    on[0, 0, 2] = 1.0                                                   # a pitch with onsets and no offset entry: skipped
    off[0, 5, 0] = 1.0                                                  # ... and one with offsets only: nothing to emit
    on[1, 3, 0] = 0.5; off[1, [3, 5], 0] = 1.0                          # an offset in the onset's own frame does not end it
    iv, pit = reshape_for_mir_eval(on, off, hop_length=4410, sample_rate=44100, min_duration=0.032)
    np.testing.assert_allclose(iv, [[0.1, 0.2], [0.4, 0.5], [0.3, 0.5]], rtol=0, atol=1e-12)
    np.testing.assert_allclose(pit, [440 * 2 ** (-68 / 12)] * 2 + [440 * 2 ** (-69 / 12)], rtol=1e-15)
    # defaults (512 / 44100 = 11.6 ms frames): a one-frame note is stretched to min_duration; "none after" = onset + int(0.032 / 0.0116) = +2 frames
    iv, pit = reshape_for_mir_eval(on, off)
    tpf = 512 / 44100
    np.testing.assert_allclose(iv, [[tpf, tpf + 0.032], [4 * tpf, 4 * tpf + 0.032], [3 * tpf, 3 * tpf + 0.032]], rtol=0, atol=1e-15)
    on2 = np.zeros((1, 8, 1)); off2 = np.zeros((1, 8, 1)); on2[0, 0, 0] = 1; off2[0, 7, 0] = 1
    iv, _ = reshape_for_mir_eval(on2, off2)
    np.testing.assert_allclose(iv, [[0.0, 7 * tpf]], atol=1e-15)         # long enough: kept as it is
    # nothing at all: the reference's placeholder note
    iv, pit = reshape_for_mir_eval(np.zeros((2, 4, 3)), np.zeros((2, 4, 3)))
    assert iv.tolist() == [[0, 0.032]] and pit.tolist() == [440.0]
    # the labels scored against themselves as train.py:194 does (offset matrix := onset matrix): a ramp of 3 non-zero frames -> 3 notes
    lab = np.zeros((1, 10, 2)); lab[0, 2:5, 1] = (0.5, 1.0, 0.5)
    iv, pit = reshape_for_mir_eval(lab, lab)
    assert len(iv) == 3 and np.allclose(iv[:, 1] - iv[:, 0], 0.032) and np.allclose(iv[:, 0], np.arange(2, 5) * tpf)
    with pytest.raises(ValueError):
        reshape_for_mir_eval(np.zeros((2, 4)), np.zeros((2, 4)))


def test_transcription_evaluate_known_answers():
    from evaluation.metrics import transcription_evaluate
    hz = lambda m: 440.0 * 2 ** ((m - 69) / 12)
    ref_i = np.array([[1.00, 1.50], [2.00, 2.40], [3.00, 4.00]]); ref_p = np.array([hz(60), hz(64), hz(67)])
    est_i = np.array([[1.04, 1.90], [2.06, 2.40], [2.95, 4.15]]); est_p = np.array([hz(60), hz(64), hz(67) * 2 ** (40 / 1200)])
    m = transcription_evaluate(ref_i, ref_p, est_i, est_p)
    # with offsets (what train.py reads): only the third note (onset -50 ms on the edge, +40 cents, offset +0.15 <= 0.2 * 1.0)
    assert m['Precision'] == pytest.approx(1 / 3) and m['Recall'] == pytest.approx(1 / 3) and m['F-measure'] == pytest.approx(1 / 3)
    assert m['Precision_no_offset'] == pytest.approx(2 / 3) and m['F-measure_no_offset'] == pytest.approx(2 / 3)
    est_p2 = est_p.copy(); est_p2[2] = hz(67) * 2 ** (60 / 1200)          # 60 cents off: a different pitch
    assert transcription_evaluate(ref_i, ref_p, est_i, est_p2)['Precision'] == 0.0
    z = transcription_evaluate(np.zeros((0, 2)), np.zeros(0), est_i, est_p)
    assert z['Precision'] == z['Recall'] == z['F-measure'] == 0.0
    # different counts: precision over the estimates, recall over the references; each estimate used once
    m = transcription_evaluate(ref_i[:1], ref_p[:1], np.array([[1.0, 1.5], [1.01, 1.5], [1.02, 1.5], [9.0, 9.5]]), np.full(4, hz(60)))
    assert m['Precision'] == 0.25 and m['Recall'] == 1.0 and m['F-measure'] == pytest.approx(0.4)
    with pytest.raises(ValueError):
        transcription_evaluate(ref_i, -ref_p, est_i, est_p)


def test_valid_metrics_pipeline_on_the_degenerate_case():
    """What the reference's default run scores: every cell of the estimate is a note of min_duration.  A reference ramp of three frames
    at one pitch then finds three matches among B * T * P estimates: recall 1, precision 3 / (B * T * P)."""
    from evaluation.metrics import reshape_for_mir_eval, transcription_evaluate
    B, T, P = 2, 16, 5
    est = np.full((B, T, P), 0.3)
    lab = np.zeros((B, T, P)); lab[1, 6:9, 2] = (0.5, 1.0, 0.5)
    ei, ep = reshape_for_mir_eval(est, est)
    ri, rp = reshape_for_mir_eval(lab, lab)
    assert len(ei) == B * T * P and len(ri) == 3
    m = transcription_evaluate(ri, rp, ei, ep)
    assert m['Recall'] == 1.0 and m['Precision'] == pytest.approx(3 / (B * T * P))


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_transcription_evaluate_against_a_brute_force_restatement_and_scipy_matching(seed):
    """An independent route to the same numbers on dense random note sets (many candidates per note, so greedy matching and maximum matching
    differ): the pairwise conditions written out as dense matrices (no sorting, no windows) and the matching size from
    scipy.sparse.csgraph.maximum_bipartite_matching instead of the module's augmenting paths."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching
    from evaluation.metrics import transcription_evaluate
    rng = np.random.default_rng(seed)
    n_ref, n_est = 300, 330
    ref_on = np.sort(rng.uniform(0, 20, n_ref)); ref_i = np.stack([ref_on, ref_on + rng.uniform(0.05, 1.0, n_ref)], 1)
    ref_p = 440.0 * 2.0 ** (rng.integers(-6, 7, n_ref) / 12.0)                    # 13 pitches: collisions in time AND pitch
    pick = rng.integers(0, n_ref, n_est)
    est_on = ref_i[pick, 0] + rng.normal(0, 0.04, n_est)
    est_i = np.stack([est_on, ref_i[pick, 1] + rng.normal(0, 0.08, n_est)], 1)
    est_i[:, 1] = np.maximum(est_i[:, 1], est_i[:, 0] + 0.01)
    est_p = ref_p[pick] * 2.0 ** (rng.normal(0, 0.25, n_est) / 12.0)              # some beyond 50 cents
    got = transcription_evaluate(ref_i, ref_p, est_i, est_p)
    d_on = np.around(np.abs(ref_i[:, None, 0] - est_i[None, :, 0]), 4)
    d_off = np.around(np.abs(ref_i[:, None, 1] - est_i[None, :, 1]), 4)
    cents = np.abs(1200.0 * np.log2(ref_p)[:, None] - 1200.0 * np.log2(est_p)[None, :])
    tol = np.maximum(0.05, 0.2 * (ref_i[:, 1] - ref_i[:, 0]))[:, None]
    for key, ok in (('_no_offset', (d_on <= 0.05) & (cents <= 50.0)), ('', (d_on <= 0.05) & (cents <= 50.0) & (d_off <= tol))):
        match = maximum_bipartite_matching(csr_matrix(ok.astype(np.int8)), perm_type='column')
        tp = int((match >= 0).sum())
        assert 0 < tp < min(n_ref, n_est)
        p, r = tp / n_est, tp / n_ref
        assert got['Precision' + key] == pytest.approx(p, abs=1e-12) and got['Recall' + key] == pytest.approx(r, abs=1e-12)
        assert got['F-measure' + key] == pytest.approx(2 * p * r / (p + r), abs=1e-12)
        # and the matching really is larger than what a greedy first-fit finds on these sets (the test would be vacuous otherwise)
        used, greedy = set(), 0
        for i in range(n_ref):
            for j in np.nonzero(ok[i])[0]:
                if j not in used:
                    used.add(j); greedy += 1
                    break
        assert greedy <= tp
