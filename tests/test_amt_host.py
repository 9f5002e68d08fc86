"""Host-side logic of model/amt.py (clip windowing, note decoding, MIDI writing) against goldens produced by the
reference's AMT.transcript / transcript_stride / mpe2note (tests/golden/make_golden.py::make_amt).  CPU only."""
import numpy as np
import torch

import util


class EchoModel:
    """Same deterministic stand-in as in make_golden.py (a function of the clip window only)."""
    def __init__(self, cfg):
        self.cfg = cfg

    def eval(self):
        return self

    def to(self, device):
        return self

    def __call__(self, spec):
        c = self.cfg
        M, T, N, nv = c['input']['margin_b'], c['input']['num_frame'], c['midi']['num_note'], c['midi']['num_velocity']
        ctr = spec[:, :N, M:M + T].transpose(1, 2)
        edge = spec[:, :N, :T].transpose(1, 2)
        vel = torch.stack([(ctr * (k + 1)).sin() for k in range(nv)], dim=-1)
        return (ctr, ctr * 0.5 + edge, ctr - 1.0, vel, None, edge, ctr + edge, ctr * 2.0, vel.flip(-1))


CFG = {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': 12, 'n_bins': 12},
       'input': {'margin_b': 2, 'margin_f': 2, 'num_frame': 8, 'min_value': -18.5},
       'midi': {'note_min': 21, 'note_max': 28, 'num_note': 8, 'num_velocity': 4}}


def _amt(batch_size):
    from model.amt import AMT
    amt = AMT(CFG, None, batch_size=batch_size)
    amt.model = EchoModel(CFG)
    amt.device = 'cpu'
    return amt


def test_transcript_windowing_matches_reference():
    g = util.golden('amt')
    for bs in (1, 3, 16):
        amt = _amt(bs)
        for n in (8, 21, 30):
            feat = g[f'tr.{n}.feature']
            outs = amt.transcript(feat)
            assert len(outs) == 8
            for i, o in enumerate(outs):
                ref = g[f'tr.{n}.out{i}']
                assert o.dtype == ref.dtype and o.shape == ref.shape
                np.testing.assert_array_equal(o, ref)
            for n_off in (0, 2, 4):
                outs = amt.transcript_stride(feat, n_off)
                for i, o in enumerate(outs):
                    ref = g[f'trs.{n}.{n_off}.out{i}']
                    assert o.dtype == ref.dtype and o.shape == ref.shape
                    np.testing.assert_array_equal(o, ref)


def test_mpe2note_matches_reference(tmp_path):
    from model.amt import AMT
    g = util.golden('amt')
    amt = AMT({'feature': {'sr': 16000, 'hop_sample': 256}, 'midi': {'note_min': 21, 'num_note': 88}}, None)
    total = 0
    for case in range(2):
        on, off, mpe, vel = (g[f'm2n.{case}.{k}'] for k in ('onset', 'offset', 'mpe', 'velocity'))
        for mv in ('ignore_zero', 'org'):
            for mo in ('shorter', 'longer', 'offset'):
                notes = amt.mpe2note(a_onset=on, a_offset=off, a_mpe=mpe, a_velocity=vel, thred_onset=0.6, thred_offset=0.55,
                                     thred_mpe=0.5, mode_velocity=mv, mode_offset=mo)
                ref = g[f'm2n.{case}.{mv}.{mo}']
                assert len(notes) == len(ref), (case, mv, mo)
                arr = np.array([[x['pitch'], x['onset'], x['offset'], x['velocity']] for x in notes]).reshape(-1, 4)
                np.testing.assert_array_equal(arr[:, [0, 3]], ref[:, [0, 3]])
                np.testing.assert_allclose(arr[:, 1:3], ref[:, 1:3], rtol=0, atol=2e-6)   # float32 vs float64 sub-frame refinement
                total += len(notes)
    assert total > 1000
    # edge cases: empty input, nothing above threshold, single frame
    assert amt.mpe2note(a_onset=np.zeros((0, 88)), a_offset=np.zeros((0, 88)), a_mpe=np.zeros((0, 88)), a_velocity=np.zeros((0, 88))) == []
    z = np.zeros((10, 88), np.float32)
    assert amt.mpe2note(a_onset=z, a_offset=z, a_mpe=z, a_velocity=z.astype(np.int8)) == []
    # MIDI writer: parse back the note-on events
    notes = [{'pitch': 60, 'onset': 0.5, 'offset': 1.0, 'velocity': 100}, {'pitch': 64, 'onset': 0.75, 'offset': 2.0, 'velocity': 64}]
    f = tmp_path / 'x.mid'
    amt.note2midi(notes, str(f))
    b = f.read_bytes()
    assert b[:4] == b'MThd' and b[14:18] == b'MTrk'
    assert bytes([0x90, 60, 100]) in b and bytes([0x90, 64, 64]) in b and bytes([0x90, 60, 0]) in b
