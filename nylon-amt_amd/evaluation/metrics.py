"""Frame-level and note-level transcription scores.

The reference does not compute these itself: `evaluation/m_mpe.py:108-110` calls `mir_eval.multipitch.evaluate` on the
thresholded mpe roll (threshold `thred_mpe`, 16 ms frames, one frequency per active MIDI pitch, `m_mpe.py:97-103`) and
`evaluation/m_transcription.py:111-123` calls `mir_eval.transcription.evaluate` on (onset, offset, pitch) lists, dropping
estimated notes with offset <= onset (`m_transcription.py:104,109`).  mir_eval is an un-vendored, unpinned dependency that is
absent here, so this module restates the PUBLISHED definitions and is **parity unpinned** at that library boundary: it is
pinned by hand-computed known answers and properties only (tests/test_metrics.py), not by mir_eval output.

Definitions restated:
  * frame level (Poliner & Ellis / mir_eval.multipitch): per frame, an estimated pitch is correct if a reference pitch lies
    within half a semitone; on a common MIDI grid that is set intersection.  Precision = TP / n_est, Recall = TP / n_ref,
    Accuracy = TP / (TP + FP + FN), all summed over frames; f1 as `m_mpe.py:166-175`.
  * note level (mir_eval.transcription, MIREX): an estimated note matches a reference note if the onsets differ by at most
    `onset_tol` (50 ms) and the pitches by at most 50 cents (same MIDI pitch here); with offsets also
    |offset difference| <= max(`offset_min_tol`, `offset_ratio` * reference duration).  Each note is used at most once:
    the score counts a MAXIMUM bipartite matching (augmenting paths), not a greedy one.
Host-side numpy: this is CPU post-processing in the reference too; nothing here runs on the GPU.
"""
from __future__ import annotations

import numpy as np


def _prf(tp, n_est, n_ref):
    p = tp / n_est if n_est > 0 else 0.0
    r = tp / n_ref if n_ref > 0 else 0.0
    f = 2.0 * p * r / (p + r) if (p + r) > 0.0 else 0.0      # m_mpe.py:166-175
    return p, r, f


def frame_metrics(ref_roll, est_roll, threshold=None):
    """ref_roll, est_roll: [n_frame, n_pitch] boolean piano rolls (or posteriors with `threshold`, `>=` as m_mpe.py:102).
    Frames beyond the shorter roll are ignored (`nframe = min(...)`, m_mpe.py:95)."""
    ref = np.asarray(ref_roll)
    est = np.asarray(est_roll)
    if ref.ndim != 2 or est.ndim != 2 or ref.shape[1] != est.shape[1]:
        raise ValueError('frame_metrics: rolls must be [n_frame, n_pitch] with the same pitch axis, got %s and %s' % (ref.shape, est.shape))
    if threshold is not None:
        ref = ref >= threshold
        est = est >= threshold
    n = min(len(ref), len(est))
    ref = ref[:n].astype(bool)
    est = est[:n].astype(bool)
    tp = int(np.logical_and(ref, est).sum())
    n_ref, n_est = int(ref.sum()), int(est.sum())
    fp, fn = n_est - tp, n_ref - tp
    p, r, f = _prf(tp, n_est, n_ref)
    acc = tp / (tp + fp + fn) if (tp + fp + fn) > 0 else 0.0
    return {'Precision': p, 'Recall': r, 'Accuracy': acc, 'f1': f, 'n_ref': n_ref, 'n_est': n_est, 'n_correct': tp}


def _max_matching(adj, n_right):
    """Size and pairs of a maximum bipartite matching; adj[i] = list of right vertices reachable from left vertex i.
    Kuhn's augmenting paths with an explicit stack (the recursive form used one Python frame per hop of a path: dense label sets could
    exceed the interpreter's recursion limit -- ADVICE r05); vertices are tried in the same order, so the matching is the same."""
    match_r = {}
    size = 0
    for root in range(len(adj)):
        if not adj[root]:
            continue
        seen = set()
        stack = [(root, iter(adj[root]))]           # the alternating path being grown: (left vertex, its untried right neighbours)
        via = []                                    # via[d] = the right vertex through which stack[d + 1] was entered
        found = False
        while stack:
            i, it = stack[-1]
            advanced = False
            for j in it:
                if j in seen:
                    continue
                seen.add(j)
                if j not in match_r:                # free right vertex: flip the whole path
                    via.append(j)
                    for d in range(len(stack)):
                        match_r[via[d]] = stack[d][0]
                    found = True
                    break
                via.append(j)
                stack.append((match_r[j], iter(adj[match_r[j]])))
                advanced = True
                break
            if found:
                break
            if not advanced:
                stack.pop()
                if via:
                    via.pop()
        if found:
            size += 1
    return size, [(match_r[j], j) for j in sorted(match_r)]


def _note_arrays(notes):
    on = np.array([float(n['onset']) for n in notes], dtype=np.float64)
    off = np.array([float(n['offset']) for n in notes], dtype=np.float64)
    pit = np.array([int(n['pitch']) for n in notes], dtype=np.int64)
    return on, off, pit


def note_metrics(ref_notes, est_notes, onset_tol=0.05, with_offset=False, offset_ratio=0.2, offset_min_tol=0.05):
    """ref_notes / est_notes: lists of dicts with 'onset', 'offset' (seconds) and 'pitch' (MIDI number) -- the format
    `AMT.mpe2note` returns.  Estimated notes with offset <= onset are dropped first (m_transcription.py:104,109)."""
    est_notes = [n for n in est_notes if float(n['offset']) - float(n['onset']) > 0.0]
    r_on, r_off, r_p = _note_arrays(ref_notes)
    e_on, e_off, e_p = _note_arrays(est_notes)
    eps = 1e-9                                     # a difference of exactly the tolerance counts (mir_eval uses <=)
    adj = []
    for i in range(len(r_on)):
        ok = (e_p == r_p[i]) & (np.abs(e_on - r_on[i]) <= onset_tol + eps)
        if with_offset:
            tol = max(offset_min_tol, offset_ratio * (r_off[i] - r_on[i]))
            ok &= np.abs(e_off - r_off[i]) <= tol + eps
        adj.append(np.nonzero(ok)[0].tolist())
    tp, pairs = _max_matching(adj, len(e_on))
    p, r, f = _prf(tp, len(e_on), len(r_on))
    return {'Precision': p, 'Recall': r, 'F-measure': f, 'n_ref': len(r_on), 'n_est': len(e_on), 'n_matched': tp, 'matching': pairs}


# ---------------------------------------------------------------------------------------------------------------------------
# The scoring branch of the reference's `training/train.py::valid(metrics=True)` -- what an unchanged `m_training.py` runs at
# its last step (7-3) by default (`m_training.py:64,466-470`).  Restated AS IT IS, degenerate or not:
# every NON-ZERO entry of the onset matrix is an onset (so every cell of a sigmoid output is one), the reference notes take the
# onset labels as their offsets too (`train.py:194`), frame time is 512 / 44100 s whatever the corpus hop is, and the clips of a
# batch are pooled on one time axis.  The scores come from `mir_eval.transcription.evaluate` (un-vendored, unpinned, absent
# here: **parity unpinned** at that boundary; published algorithm restated, pinned by hand-computed answers in tests/test_metrics.py).
# ---------------------------------------------------------------------------------------------------------------------------

def reshape_for_mir_eval(onset_matrix, offset_matrix, hop_length=512, sample_rate=44100, min_duration=0.032):
    """`training/train.py:9-57`: [batch, frame, pitch] matrices -> (intervals [n, 2] seconds, pitches [n] Hz).

    Per (clip, pitch) with at least one non-zero onset AND one non-zero offset entry: every non-zero onset frame becomes a note
    ending at the first non-zero offset frame strictly after it (none: onset + max(1, int(min_duration / frame time)) frames),
    stretched to `min_duration`; pitch index p sounds at 440 * 2^((p - 69) / 12) Hz.  Notes are emitted clip-major, pitch, onset
    order as the reference's loops do.  An empty result is the reference's placeholder note ([[0, min_duration]], [440.0])."""
    on = np.asarray(onset_matrix) != 0
    off = np.asarray(offset_matrix) != 0
    if on.ndim != 3 or on.shape != off.shape:
        raise ValueError('reshape_for_mir_eval: [batch, frame, pitch] matrices of one shape expected, got %s and %s' % (on.shape, off.shape))
    tpf = hop_length / sample_rate
    n_b, n_t, n_p = on.shape
    fill = max(1, int(min_duration / tpf))
    # first non-zero offset frame strictly after frame t, per (clip, pitch): a reversed running minimum
    idx = np.where(off, np.arange(n_t, dtype=np.int64)[None, :, None], np.int64(n_t))       # n_t = "none"
    nxt = np.minimum.accumulate(idx[:, ::-1, :], axis=1)[:, ::-1, :]                        # first offset frame >= t
    after = np.concatenate([nxt[:, 1:, :], np.full((n_b, 1, n_p), n_t, dtype=np.int64)], axis=1)   # first offset frame > t
    live = on & off.any(axis=1, keepdims=True)                  # `len(offset_frames) == 0: continue`
    b, p, t = np.nonzero(live.transpose(0, 2, 1))               # clip-major, then pitch, then onset frame
    if len(t) == 0:
        return np.array([[0, min_duration]]), np.array([440.0])
    end = after[b, t, p]
    end = np.where(end >= n_t, t + fill, end)
    end = np.where(end <= t, t + fill, end)
    onset_time = t * tpf
    offset_time = end * tpf
    offset_time = np.where(offset_time - onset_time < min_duration, onset_time + min_duration, offset_time)
    intervals = np.stack([onset_time, offset_time], axis=1)
    pitches = 440.0 * (2.0 ** ((p - 69) / 12.0))
    keep = (intervals[:, 1] - intervals[:, 0]) > 0
    return intervals[keep], pitches[keep]


_N_DECIMALS = 4           # mir_eval.transcription rounds the distances to four decimals before comparing with a tolerance


def transcription_evaluate(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05, pitch_tolerance=50.0,
                           offset_ratio=0.2, offset_min_tolerance=0.05):
    """What `training/train.py:196-199` reads from `mir_eval.transcription.evaluate`: 'Precision', 'Recall', 'F-measure' of
    `precision_recall_f1_overlap` with its defaults, i.e. WITH offsets: an estimated note matches a reference note when the
    onsets are within 50 ms, the pitches within 50 cents and the offsets within max(50 ms, 0.2 x reference duration); distances
    are rounded to four decimals; a maximum bipartite matching is counted; either list empty -> zeros.  Also returns the
    onset-only scores under mir_eval's names ('Precision_no_offset', ...)."""
    ref_i = np.asarray(ref_intervals, dtype=np.float64).reshape(-1, 2)
    est_i = np.asarray(est_intervals, dtype=np.float64).reshape(-1, 2)
    ref_p = np.asarray(ref_pitches, dtype=np.float64).reshape(-1)
    est_p = np.asarray(est_pitches, dtype=np.float64).reshape(-1)
    if len(ref_i) != len(ref_p) or len(est_i) != len(est_p):
        raise ValueError('transcription_evaluate: one pitch per interval expected')
    if (ref_p <= 0).any() or (est_p <= 0).any():
        raise ValueError('transcription_evaluate: pitches are frequencies in Hz and must be positive')
    out = {}
    for key, with_offset in (('', True), ('_no_offset', False)):
        tp = 0
        if len(ref_i) and len(est_i):
            # candidates per reference note, found among the estimates sorted by onset (the default run has ~1e5 of them)
            order = np.argsort(est_i[:, 0], kind='stable')
            e_on, e_off = est_i[order, 0], est_i[order, 1]
            e_cents = 1200.0 * np.log2(est_p[order])
            r_cents = 1200.0 * np.log2(ref_p)
            slack = onset_tolerance + 1e-3
            lo = np.searchsorted(e_on, ref_i[:, 0] - slack, side='left')
            hi = np.searchsorted(e_on, ref_i[:, 0] + slack, side='right')
            adj = []
            for i in range(len(ref_i)):
                s = slice(lo[i], hi[i])
                ok = np.around(np.abs(e_on[s] - ref_i[i, 0]), _N_DECIMALS) <= onset_tolerance
                ok &= np.abs(e_cents[s] - r_cents[i]) <= pitch_tolerance
                if with_offset:
                    tol = max(offset_min_tolerance, offset_ratio * (ref_i[i, 1] - ref_i[i, 0]))
                    ok &= np.around(np.abs(e_off[s] - ref_i[i, 1]), _N_DECIMALS) <= tol
                adj.append((lo[i] + np.nonzero(ok)[0]).tolist())
            tp, _ = _max_matching(adj, len(est_i))
        p, r, f = _prf(tp, len(est_i), len(ref_i))
        out['Precision' + key], out['Recall' + key], out['F-measure' + key] = p, r, f
    return out
