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
    """Size and pairs of a maximum bipartite matching; adj[i] = list of right vertices reachable from left vertex i."""
    match_r = [-1] * n_right

    def try_left(i, seen):
        for j in adj[i]:
            if not seen[j]:
                seen[j] = True
                if match_r[j] < 0 or try_left(match_r[j], seen):
                    match_r[j] = i
                    return True
        return False

    size = 0
    for i in range(len(adj)):
        if adj[i] and try_left(i, [False] * n_right):
            size += 1
    return size, [(match_r[j], j) for j in range(n_right) if match_r[j] >= 0]


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
