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


# ---------------------------------------------------------------------------------------------------------------------
# WAV loader and resampler of wav2feature (reference: torchaudio.load + transforms.Resample, amt.py:55-58; torchaudio is
# not in this image, so the resampler is a restatement of its published algorithm (oracle.hftt_oracle.resample; the product runs the HIP kernel
# hftt_resample, held against that restatement in tests/test_small_kernels_gpu.py) -- checked here against analytic
# signals and against scipy's polyphase resampler as an independent implementation).
# ---------------------------------------------------------------------------------------------------------------------
import math
import pytest


@pytest.mark.parametrize('sr_in', [44100, 48000, 22050, 8000])
def test_resample_reproduces_a_band_limited_signal(sr_in):
    from util import O
    _resample = O.resample
    sr_out, dur = 16000, 0.25
    n = int(sr_in * dur)
    f1, f2 = 440.0, 2500.0                                    # both below every Nyquist involved
    t_in = torch.arange(n, dtype=torch.float64) / sr_in
    x = (0.6 * torch.sin(2 * math.pi * f1 * t_in) + 0.3 * torch.cos(2 * math.pi * f2 * t_in)).float()
    y = _resample(x, sr_in, sr_out)
    g = math.gcd(sr_in, sr_out)
    assert y.numel() == math.ceil((sr_out // g) * n / (sr_in // g))               # torchaudio's output length
    t_out = torch.arange(y.numel(), dtype=torch.float64) / sr_out
    ref = 0.6 * torch.sin(2 * math.pi * f1 * t_out) + 0.3 * torch.cos(2 * math.pi * f2 * t_out)
    edge = 64                                                                      # the filter's support at the borders sees zero padding
    err = (y.double() - ref)[edge:-edge].abs().max().item()
    assert err < 5e-3, err
    from scipy.signal import resample_poly
    z = resample_poly(x.double().numpy(), sr_out // g, sr_in // g)
    m = min(len(z), y.numel())
    assert np.abs(z[edge:m - edge] - y.double().numpy()[edge:m - edge]).max() < 2e-2


def test_resample_removes_what_the_new_rate_cannot_carry():
    from util import O
    _resample = O.resample
    sr_in, sr_out = 44100, 16000
    t = torch.arange(sr_in // 4, dtype=torch.float64) / sr_in
    tone = torch.sin(2 * math.pi * 11000.0 * t).float()                            # above the new Nyquist (8 kHz)
    y = _resample(tone, sr_in, sr_out)
    assert y[64:-64].abs().max().item() < 2e-2
    dc = _resample(torch.ones(sr_in // 4), sr_in, sr_out)
    assert (dc[64:-64] - 1.0).abs().max().item() < 2e-3                            # unit gain in the pass band


@pytest.mark.parametrize('dtype,scale,offset', [(np.int16, 32768.0, 0.0), (np.int32, 2147483648.0, 0.0), (np.uint8, 128.0, 128.0), (np.float32, 1.0, 0.0)])
def test_wav_loader_scales_and_orders_channels(tmp_path, dtype, scale, offset):
    from scipy.io import wavfile
    from model.amt import _load_wav
    rng = np.random.default_rng(3)
    want = rng.uniform(-0.9, 0.9, size=(1000, 2)).astype(np.float32)               # [frames, channels] as wavfile stores it
    raw = want if dtype == np.float32 else np.round(want * scale + offset).astype(dtype)
    p = tmp_path / 'a.wav'
    wavfile.write(str(p), 22050, raw)
    x, sr = _load_wav(str(p))
    assert sr == 22050 and x.shape == (2, 1000) and x.dtype == np.float32 and x.flags['C_CONTIGUOUS']
    assert np.abs(x.T - want).max() <= (1.0 / scale if dtype != np.float32 else 0.0) + 1e-7
    mono = tmp_path / 'm.wav'
    wavfile.write(str(mono), 16000, raw[:, 0].copy())
    x1, sr1 = _load_wav(str(mono))
    assert sr1 == 16000 and x1.shape == (1, 1000)
