"""CPU oracle for the hFT-Transformer hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a functional, fp32, CPU restatement of the reference algorithm.  It is
*never* imported by the product package (``nylon-amt_amd/``); only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may use it,
and there only as the checker / the CPU baseline, never as the thing shipped.

Parity pin: the model/loss/Adam part is pinned against golden vectors produced by
importing the reference's ``hftt_code/model/model_spec2midi.py`` in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every one of them).  The log-mel front end
restates torchaudio's published MelSpectrogram algorithm (torchaudio is an
un-vendored, unpinned dependency of the reference: ``model/amt.py:6``); no reference
fixture exists for it, so that part is "parity unpinned" (self-checked against
``torch.stft`` only).

Each function cites the reference lines it follows (paths relative to
``/root/reference/hftt_code``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------
@dataclass(frozen=True)
class HfttConfig:
    """Shape constants; defaults = corpus/config.json + paper-size flags (m_training.py:55-60 help strings)."""
    n_margin: int = 32
    n_frame: int = 128
    n_bin: int = 256
    cnn_channel: int = 4
    cnn_kernel: int = 5
    hid_dim: int = 256
    pf_dim: int = 512
    enc_layer: int = 3
    dec_layer: int = 3
    enc_head: int = 4
    dec_head: int = 4
    n_note: int = 88
    n_velocity: int = 128

    @property
    def n_proc(self):          # model_spec2midi.py:52
        return 2 * self.n_margin + 1

    @property
    def cnn_dim(self):         # model_spec2midi.py:53
        return self.cnn_channel * (self.n_proc - (self.cnn_kernel - 1))

    def as_dict(self):
        return asdict(self)


MICRO = HfttConfig(n_margin=4, n_frame=8, n_bin=16, cnn_channel=4, cnn_kernel=5, hid_dim=16, pf_dim=32,
                   enc_layer=2, dec_layer=2, enc_head=2, dec_head=2, n_note=6, n_velocity=8)
TINY = HfttConfig(hid_dim=64, pf_dim=128, enc_layer=2, dec_layer=2, enc_head=2, dec_head=2)   # m_training.py:55-60 defaults
PAPER = HfttConfig()


# ----------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------
def _drop(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0.0) else x


def mha(sd, pre, q_in, k_in, v_in, n_heads, p=0.0, training=False):
    """MultiHeadAttentionLayer.forward, model_spec2midi.py:322-360.

    Returns (projected output, pre-dropout softmax probabilities)."""
    bsz, lq, d = q_in.shape
    lk = k_in.shape[1]
    dh = d // n_heads
    q = F.linear(q_in, sd[pre + 'fc_q.weight'], sd[pre + 'fc_q.bias'])            # :328
    k = F.linear(k_in, sd[pre + 'fc_k.weight'], sd[pre + 'fc_k.bias'])            # :329
    v = F.linear(v_in, sd[pre + 'fc_v.weight'], sd[pre + 'fc_v.bias'])            # :330
    q = q.view(bsz, lq, n_heads, dh).transpose(1, 2)                               # :335
    k = k.view(bsz, lk, n_heads, dh).transpose(1, 2)
    v = v.view(bsz, lk, n_heads, dh).transpose(1, 2)
    energy = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)                  # :342  (scale = sqrt(head_dim), :320)
    prob = torch.softmax(energy, dim=-1)                                           # :345
    ctx = torch.matmul(_drop(prob, p, training), v)                                # :348
    ctx = ctx.transpose(1, 2).contiguous().view(bsz, lq, d)                        # :351-354
    out = F.linear(ctx, sd[pre + 'fc_o.weight'], sd[pre + 'fc_o.bias'])            # :357
    return out, prob


def ffn(sd, pre, x, p=0.0, training=False):
    """PositionwiseFeedforwardLayer.forward, model_spec2midi.py:369-378."""
    h = _drop(torch.relu(F.linear(x, sd[pre + 'fc_1.weight'], sd[pre + 'fc_1.bias'])), p, training)
    return F.linear(h, sd[pre + 'fc_2.weight'], sd[pre + 'fc_2.bias'])


def _ln(sd, pre, x):
    # one nn.LayerNorm per layer, shared by every post-norm of that layer (model_spec2midi.py:225,250,277)
    return F.layer_norm(x, (x.shape[-1],), sd[pre + 'layer_norm.weight'], sd[pre + 'layer_norm.bias'], 1e-5)


def encoder_layer(sd, pre, x, n_heads, p=0.0, training=False):
    """EncoderLayer.forward, model_spec2midi.py:230-245 (freq self-attention and time self-attention)."""
    a, _ = mha(sd, pre + 'self_attention.', x, x, x, n_heads, p, training)
    x = _ln(sd, pre, x + _drop(a, p, training))                                    # :236
    f = ffn(sd, pre + 'positionwise_feedforward.', x, p, training)
    return _ln(sd, pre, x + _drop(f, p, training))                                 # :242


def decoder_layer_zero(sd, pre, enc, trg, n_heads, p=0.0, training=False):
    """DecoderLayer_Zero.forward, model_spec2midi.py:255-272."""
    a, prob = mha(sd, pre + 'encoder_attention.', trg, enc, enc, n_heads, p, training)
    trg = _ln(sd, pre, trg + _drop(a, p, training))
    f = ffn(sd, pre + 'positionwise_feedforward.', trg, p, training)
    return _ln(sd, pre, trg + _drop(f, p, training)), prob


def decoder_layer(sd, pre, enc, trg, n_heads, p=0.0, training=False):
    """DecoderLayer.forward, model_spec2midi.py:283-306."""
    a, _ = mha(sd, pre + 'self_attention.', trg, trg, trg, n_heads, p, training)
    trg = _ln(sd, pre, trg + _drop(a, p, training))
    a, prob = mha(sd, pre + 'encoder_attention.', trg, enc, enc, n_heads, p, training)
    trg = _ln(sd, pre, trg + _drop(a, p, training))
    f = ffn(sd, pre + 'positionwise_feedforward.', trg, p, training)
    return _ln(sd, pre, trg + _drop(f, p, training)), prob


# ----------------------------------------------------------------------------------
# encoder / decoder / model
# ----------------------------------------------------------------------------------
def encoder_forward(sd, spec_in, cfg: HfttConfig, p=0.0, training=False):
    """Encoder_SPEC2MIDI.forward, model_spec2midi.py:60-106.  spec_in [B, n_bin, M+T+M] -> [B, T, n_bin, d]."""
    pre = 'encoder_spec2midi.'
    bsz = spec_in.shape[0]
    T, Fq, d = cfg.n_frame, cfg.n_bin, cfg.hid_dim
    win = spec_in.unfold(2, cfg.n_proc, 1).permute(0, 2, 1, 3).contiguous()        # :65  [B,T,F,n_proc]
    win = win.reshape(bsz * T, 1, Fq, cfg.n_proc)                                  # :70
    cnn = F.conv2d(win, sd[pre + 'conv.weight'], sd[pre + 'conv.bias'])            # :73  [B*T,C,F,n_proc-k+1]
    cnn = cnn.permute(0, 2, 1, 3).contiguous().reshape(bsz * T, Fq, cfg.cnn_dim)   # :73,80  (feature = c*61+w)
    tok = F.linear(cnn, sd[pre + 'tok_embedding_freq.weight'], sd[pre + 'tok_embedding_freq.bias'])   # :85
    pos = sd[pre + 'pos_embedding_freq.weight'][:Fq]                               # :90,95 (arange positions)
    x = _drop(tok * math.sqrt(d) + pos.unsqueeze(0), p, training)                  # :95   scale_freq = sqrt(hid_dim), :58
    for i in range(cfg.enc_layer):                                                 # :100
        x = encoder_layer(sd, f'{pre}layers_freq.{i}.', x, cfg.enc_head, p, training)
    return x.reshape(bsz, T, Fq, d)                                                # :102


def decoder_forward(sd, enc, cfg: HfttConfig, p=0.0, training=False):
    """Decoder_SPEC2MIDI.forward, model_spec2midi.py:145-216."""
    pre = 'decoder_spec2midi.'
    bsz = enc.shape[0]
    T, Fq, N, V, d = cfg.n_frame, cfg.n_bin, cfg.n_note, cfg.n_velocity, cfg.hid_dim
    enc = enc.reshape(bsz * T, Fq, d)                                              # :147
    q0 = sd[pre + 'pos_embedding_freq.weight'][:N].unsqueeze(0).expand(bsz * T, N, d)   # :154-155
    x, prob = decoder_layer_zero(sd, pre + 'layer_zero_freq.', enc, q0, cfg.dec_head, p, training)   # :161
    for i in range(cfg.dec_layer - 1):                                             # :162
        x, prob = decoder_layer(sd, f'{pre}layers_freq.{i}.', enc, x, cfg.dec_head, p, training)
    attention = prob.reshape(bsz, T, prob.shape[1], prob.shape[2], prob.shape[3])  # :164-165

    def heads(z, tag):
        on = torch.sigmoid(F.linear(z, sd[f'{pre}fc_onset_{tag}.weight'], sd[f'{pre}fc_onset_{tag}.bias']).squeeze(-1))
        of = torch.sigmoid(F.linear(z, sd[f'{pre}fc_offset_{tag}.weight'], sd[f'{pre}fc_offset_{tag}.bias']).squeeze(-1))
        mp = torch.sigmoid(F.linear(z, sd[f'{pre}fc_mpe_{tag}.weight'], sd[f'{pre}fc_mpe_{tag}.bias']).squeeze(-1))
        ve = F.linear(z, sd[f'{pre}fc_velocity_{tag}.weight'], sd[f'{pre}fc_velocity_{tag}.bias'])
        return on, of, mp, ve

    on_a, of_a, mp_a, ve_a = heads(x, 'freq')                                      # :172-175
    on_a, of_a, mp_a = (t.reshape(bsz, T, N) for t in (on_a, of_a, mp_a))
    ve_a = ve_a.reshape(bsz, T, N, V)

    y = x.reshape(bsz, T, N, d).permute(0, 2, 1, 3).contiguous().reshape(bsz * N, T, d)   # :189
    pos_t = sd[pre + 'pos_embedding_time.weight'][:T]                              # :190
    y = _drop(y * math.sqrt(d) + pos_t.unsqueeze(0), p, training)                  # :191  scale_time, :135
    for i in range(cfg.dec_layer):                                                 # :197
        y = encoder_layer(sd, f'{pre}layers_time.{i}.', y, cfg.dec_head, p, training)
    on_b, of_b, mp_b, ve_b = heads(y, 'time')                                      # :203-206
    on_b, of_b, mp_b = (t.reshape(bsz, N, T).permute(0, 2, 1).contiguous() for t in (on_b, of_b, mp_b))
    ve_b = ve_b.reshape(bsz, N, T, V).permute(0, 2, 1, 3).contiguous()
    return on_a, of_a, mp_a, ve_a, attention, on_b, of_b, mp_b, ve_b               # :216


def model_forward(sd, spec_in, cfg: HfttConfig, p=0.0, training=False):
    """Model_SPEC2MIDI.forward, model_spec2midi.py:15-35."""
    return decoder_forward(sd, encoder_forward(sd, spec_in, cfg, p, training), cfg, p, training)


# ----------------------------------------------------------------------------------
# loss (training/train.py:106-153), init (m_training.py:31-33), Adam (m_training.py:146)
# ----------------------------------------------------------------------------------
def bce_mean(prob, target):
    """nn.BCELoss(reduction='mean'): log terms clamped at -100 (torch semantics)."""
    lp = torch.clamp(torch.log(prob), min=-100.0)
    l1p = torch.clamp(torch.log1p(-prob), min=-100.0)
    return -(target * lp + (1.0 - target) * l1p).mean()


def spec2midi_loss(outputs, label_onset, label_offset, label_mpe, label_velocity, weight_A=1.0, weight_B=1.0):
    """loss = wA*(onset+offset+mpe+velocity)_A + wB*(...)_B, train.py:141-153."""
    on_a, of_a, mp_a, ve_a, _att, on_b, of_b, mp_b, ve_b = outputs
    lo, lf, lm = (t.reshape(-1).float() for t in (label_onset, label_offset, label_mpe))   # :129-131
    lv = label_velocity.reshape(-1).long()                                                 # :132
    def side(on, of, mp, ve):
        return (bce_mean(on.reshape(-1), lo) + bce_mean(of.reshape(-1), lf) + bce_mean(mp.reshape(-1), lm)
                + F.cross_entropy(ve.reshape(-1, ve.shape[-1]), lv))
    return weight_A * side(on_a, of_a, mp_a, ve_a) + weight_B * side(on_b, of_b, mp_b, ve_b)


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.Adam step (defaults of m_training.py:146; no weight decay, no amsgrad). In place."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for p_, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        m.mul_(beta1).add_(g, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p_.addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------------
# synthetic deterministic inputs (exactly reproducible on any platform: integer hash -> dyadic floats)
# ----------------------------------------------------------------------------------
def _hash_u32(idx: np.ndarray, salt: int) -> np.ndarray:
    x = (idx.astype(np.uint64) * np.uint64(2654435761) + np.uint64((salt * 0x9E3779B9) & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def synth_spec(bsz: int, cfg: HfttConfig, salt: int = 1) -> torch.Tensor:
    """Log-mel-range clip batch [B, n_bin, M+T+M] in [-18, 6): k*(24/4096)-18, k from an integer hash."""
    n = bsz * cfg.n_bin * (cfg.n_frame + 2 * cfg.n_margin)
    k = (_hash_u32(np.arange(n), salt) >> np.uint32(20)).astype(np.float32)        # 12 bits
    x = k * np.float32(24.0 / 4096.0) - np.float32(18.0)
    return torch.from_numpy(x.reshape(bsz, cfg.n_bin, cfg.n_frame + 2 * cfg.n_margin))


def synth_labels(bsz: int, cfg: HfttConfig, salt: int = 2):
    """Labels with the dataset.py:59-71 contract: onset/offset fp32 in [0,1], mpe {0,1} float, velocity int64."""
    n = bsz * cfg.n_frame * cfg.n_note
    h = _hash_u32(np.arange(n), salt)
    shape = (bsz, cfg.n_frame, cfg.n_note)
    onset = ((h & np.uint32(7)) == 0) * (((h >> np.uint32(3)) & np.uint32(3)).astype(np.float32) + 1.0) / 4.0
    offset = (((h >> np.uint32(5)) & np.uint32(7)) == 0) * (((h >> np.uint32(8)) & np.uint32(3)).astype(np.float32) + 1.0) / 4.0
    mpe = (((h >> np.uint32(10)) & np.uint32(3)) == 0).astype(np.float32)
    vel = (mpe * ((h >> np.uint32(12)) % np.uint32(cfg.n_velocity))).astype(np.int64)
    return (torch.from_numpy(onset.astype(np.float32).reshape(shape)), torch.from_numpy(offset.astype(np.float32).reshape(shape)),
            torch.from_numpy(mpe.reshape(shape)), torch.from_numpy(vel.reshape(shape)))


# ----------------------------------------------------------------------------------
# clip windowing of AMT.transcript / transcript_stride (model/amt.py:66-176), with any forward callable
# ----------------------------------------------------------------------------------
def transcript(a_feature, forward, cfg: HfttConfig, min_value: float):
    """AMT.transcript (mode='combination'), amt.py:66-118.  forward(spec[1,n_bin,192]) -> 9-tuple."""
    a_feature = np.asarray(a_feature, dtype=np.float32)
    T, M = cfg.n_frame, cfg.n_margin
    n = a_feature.shape[0]
    len_s = int(np.ceil(n / T) * T) - n                                            # :70
    a_input = np.concatenate([np.full((M, cfg.n_bin), min_value, np.float32), a_feature,
                              np.full((len_s + M, cfg.n_bin), min_value, np.float32)], axis=0)   # :69-72
    outs = [np.zeros((n + len_s, cfg.n_note), np.float32) for _ in range(3)] + [np.zeros((n + len_s, cfg.n_note), np.int8)]
    outs = outs + [o.copy() for o in outs]
    for i in range(0, n, T):                                                       # :88
        spec = torch.from_numpy(a_input[i:i + M + T + M]).T.unsqueeze(0)           # :89
        with torch.no_grad():
            o = forward(spec)
        sel = [o[0], o[1], o[2], o[3], o[5], o[6], o[7], o[8]]
        for j, t in enumerate(sel):
            t = t.squeeze(0)
            if j % 4 == 3:
                t = t.argmax(2)                                                    # :107,113
            outs[j][i:i + T] = t.cpu().numpy()
    return tuple(outs)


def transcript_stride(a_feature, n_offset, forward, cfg: HfttConfig, min_value: float):
    """AMT.transcript_stride (mode='combination'), amt.py:121-176."""
    a_feature = np.asarray(a_feature, dtype=np.float32)
    T, M = cfg.n_frame, cfg.n_margin
    half = T // 2                                                                  # :125
    n = a_feature.shape[0]
    tmp_len = n + 2 * M + half                                                     # :127
    len_s = int(np.ceil(tmp_len / half) * half) - tmp_len                          # :128
    a_input = np.concatenate([np.full((M + n_offset, cfg.n_bin), min_value, np.float32), a_feature,
                              np.full((len_s + M + (half - n_offset), cfg.n_bin), min_value, np.float32)], axis=0)
    outs = [np.zeros((n + len_s, cfg.n_note), np.float32) for _ in range(3)] + [np.zeros((n + len_s, cfg.n_note), np.int8)]
    outs = outs + [o.copy() for o in outs]
    for i in range(0, n, half):                                                    # :146
        spec = torch.from_numpy(a_input[i:i + M + T + M]).T.unsqueeze(0)           # :147
        with torch.no_grad():
            o = forward(spec)
        sel = [o[0], o[1], o[2], o[3], o[5], o[6], o[7], o[8]]
        for j, t in enumerate(sel):
            t = t.squeeze(0)[n_offset:n_offset + half]                             # :162-171
            if j % 4 == 3:
                t = t.argmax(2)
            outs[j][i:i + half] = t.cpu().numpy()
    return tuple(outs)


# ----------------------------------------------------------------------------------
# log-mel front end (model/amt.py:55-63) -- restates torchaudio.transforms.MelSpectrogram; PARITY UNPINNED
# ----------------------------------------------------------------------------------
def mel_filterbank(sr=16000, n_fft=2048, n_mels=256) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(n_freqs=n_fft//2+1, f_min=0, f_max=sr/2, norm='slaney', mel_scale='htk').

    Returns fb [n_freqs, n_mels] float32 (triangles on linspace(0, sr/2, n_freqs), area-normalised)."""
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sr // 2, n_freqs)
    m_max = 2595.0 * math.log10(1.0 + (sr / 2.0) / 700.0)
    m_pts = torch.linspace(0.0, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return fb * enorm.unsqueeze(0)


def logmel(wave_mono_16k: torch.Tensor, sr=16000, n_fft=2048, hop=256, n_mels=256, log_offset=1e-8) -> torch.Tensor:
    """AMT.wav2feature after resampling (amt.py:58-61): power mel spectrogram (hann periodic, center, zero pad)
    -> log(mel + 1e-8) -> [n_frames, n_mels], n_frames = 1 + len//hop."""
    win = torch.hann_window(n_fft, periodic=True)
    st = torch.stft(wave_mono_16k.float(), n_fft, hop_length=hop, win_length=n_fft, window=win, center=True,
                    pad_mode='constant', normalized=False, onesided=True, return_complex=True)
    power = st.real ** 2 + st.imag ** 2                                            # power=2.0   [n_freqs, n_frames]
    mel = torch.matmul(power.transpose(-1, -2), mel_filterbank(sr, n_fft, n_mels))  # [n_frames, n_mels]
    return torch.log(mel + log_offset)


def logmel_dft(wave_mono_16k: torch.Tensor, n_fft=2048, hop=256, n_mels=256, log_offset=1e-8, sr=16000) -> torch.Tensor:
    """Same as :func:`logmel` but by explicit framing + real DFT matrices in float64 (no torch.stft): the
    independent self-check of the framing convention (center=True, zero padding n_fft/2 each side)."""
    x = F.pad(wave_mono_16k.double(), (n_fft // 2, n_fft // 2))
    n_frames = 1 + (x.numel() - n_fft) // hop
    frames = x.unfold(0, n_fft, hop)[:n_frames] * torch.hann_window(n_fft, periodic=True, dtype=torch.float64)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64).unsqueeze(1)
    n = torch.arange(n_fft, dtype=torch.float64).unsqueeze(0)
    ang = 2.0 * math.pi * k * n / n_fft
    re = frames @ torch.cos(ang).T
    im = frames @ torch.sin(ang).T
    power = re * re + im * im
    return torch.log(power @ mel_filterbank(sr, n_fft, n_mels).double() + log_offset).float()


def resample(wave: torch.Tensor, sr_in: int, sr_out: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """model/amt.py:57-58: torchaudio.transforms.Resample(sr, 16000) with its defaults, restated from the published algorithm (Hann-windowed
    sinc interpolation, torchaudio.functional.resample: `_get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`); float64 on the CPU.
    torchaudio is un-vendored, unpinned and absent: PARITY UNPINNED at that boundary -- this restatement is what the HIP kernel
    (hftt_resample) is held against, and it is itself checked on band-limited signals (tests/test_amt_host.py)."""
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, :] / orig
    t = (torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx) * base
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    x = torch.nn.functional.pad(wave.double()[None, None, :], (width, width + orig))
    y = torch.nn.functional.conv1d(x, kern[:, None, :], stride=orig)      # [1, new, frames]
    y = y.transpose(1, 2).reshape(-1)
    target = int(math.ceil(new * wave.numel() / orig))
    return y[:target].float()
