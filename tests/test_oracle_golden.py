"""Pin the CPU oracle (oracle/hftt_oracle.py) against the golden vectors generated from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import util
from util import O


def _sd(g, prefix='sd.'):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def test_micro_forward_loss_grads_adam():
    g = util.golden('micro')
    cfg = util.cfg_from_golden(g)
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(g).items()}
    x = torch.from_numpy(g['input'])
    labels = [torch.from_numpy(g[n]) for n in ('label_onset', 'label_offset', 'label_mpe', 'label_velocity')]
    out = O.model_forward(sd, x, cfg)
    for n, t in zip(util.OUT_NAMES, out):
        assert t.shape == g['out.' + n].shape
        assert util.max_err(t, torch.from_numpy(g['out.' + n])) < 2e-6, n
    loss = O.spec2midi_loss(out, *labels, float(g['weight_A']), float(g['weight_B']))
    assert abs(loss.item() - float(g['loss'])) < 1e-5
    loss.backward()
    for k, v in sd.items():
        ref = torch.from_numpy(g['grad.' + k])
        assert util.max_err(v.grad, ref) < 1e-6 + 1e-4 * ref.abs().max().item(), k
    # one Adam step (m_training.py:146 defaults)
    names = list(sd.keys())
    params = [sd[k].detach().clone() for k in names]
    grads = [sd[k].grad for k in names]
    m = [torch.zeros_like(p) for p in params]; v = [torch.zeros_like(p) for p in params]
    O.adam_step(params, grads, m, v, 1, lr=1e-4)
    for k, p, gr in zip(names, params, grads):
        if gr.abs().max().item() < 1e-7:
            continue     # fc_k.bias: its true gradient is 0 (softmax is invariant to a key bias); Adam amplifies rounding noise
        assert util.max_err(p, torch.from_numpy(g['adam1.' + k])) < 2e-7, k


def _check_big(name, with_grads):
    g = util.golden(name)
    cfg = util.cfg_from_golden(g)
    seed, bsz = int(g['seed']), int(g['bsz'])
    model = util.build_model(cfg, seed)           # our boundary module: same RNG stream as the reference's constructor
    util.perturb(model, seed + 1)
    sd = util.sd_cpu(model)
    for k, v in sd.items():
        ref = g['sdsum.' + k]
        assert abs(v.double().sum().item() - ref[0]) <= 1e-9 * max(1.0, abs(ref[1])), k
        assert abs(v.double().abs().sum().item() - ref[1]) <= 1e-9 * max(1.0, abs(ref[1])), k
    x = O.synth_spec(bsz, cfg, salt=seed)
    labels = O.synth_labels(bsz, cfg, salt=seed + 7)
    if with_grads:
        sd = {k: v.requires_grad_(True) for k, v in sd.items()}
        out = O.model_forward(sd, x, cfg)
    else:
        with torch.no_grad():
            out = O.model_forward(sd, x, cfg)
    for n, t in zip(util.OUT_NAMES, out):
        st = int(g['out.' + n + '.stride'])
        ref = torch.from_numpy(g['out.' + n + '.sample'])
        tol = 5e-6 * max(1.0, float(g['out.' + n + '.stats'][1]))
        assert util.max_err(t.reshape(-1)[::st], ref) < tol, n
    loss = O.spec2midi_loss(out, *labels)
    assert abs(loss.item() - float(g['loss'])) < 2e-5
    if with_grads:
        loss.backward()
        for k, v in sd.items():
            gs = g['grad.' + k + '.stats']
            assert abs(v.grad.double().norm().item() - gs[2]) < 1e-4 * gs[2] + 1e-7, k


def test_tiny_b2_forward_loss_grads():
    _check_big('tiny_b2', True)


def test_paper_b1_forward_loss():
    _check_big('paper_b1', False)


def test_logmel_self_consistency():
    """Front end is 'parity unpinned' (torchaudio absent/unpinned): check the restatement against an independent
    float64 framing + DFT implementation and basic invariants."""
    g = torch.Generator().manual_seed(0)
    n = 16000 + 77
    t = torch.arange(n) / 16000.0
    wave = 0.5 * torch.sin(2 * np.pi * 440.0 * t) + 0.01 * torch.randn(n, generator=g)
    a = O.logmel(wave)
    b = O.logmel_dft(wave)
    assert a.shape == (1 + n // 256, 256)
    assert util.max_err(a, b) < 2e-3
    fb = O.mel_filterbank()
    assert fb.shape == (1025, 256)
    assert int((fb > 0).sum()) == 2036            # 2,036 non-zeros
    silence = O.logmel(torch.zeros(4096))
    assert util.max_err(silence, torch.full_like(silence, float(np.log(np.float32(1e-8))))) < 1e-5
    peak_bin = a[10].argmax().item()
    f_pts = 700.0 * (10.0 ** (torch.linspace(0, 2595.0 * np.log10(1 + 8000 / 700.0), 258) / 2595.0) - 1.0)
    assert f_pts[peak_bin] <= 440.0 <= f_pts[peak_bin + 2]


def test_transcript_windowing_identity_model():
    """amt.py:66-176 windowing restatement with a model that echoes (a function of) its input frames."""
    cfg = O.HfttConfig(n_margin=2, n_frame=8, n_bin=4, n_note=4, n_velocity=4, hid_dim=16, pf_dim=16)
    n = 21
    feat = np.arange(n * cfg.n_bin, dtype=np.float32).reshape(n, cfg.n_bin)

    def fwd(spec):     # spec [1, n_bin, M+T+M]; returns frame-aligned copies of the centre frames
        c = spec[0, :, cfg.n_margin:cfg.n_margin + cfg.n_frame].T.unsqueeze(0)      # [1,T,n_bin] == [1,T,n_note]
        vel = torch.zeros(1, cfg.n_frame, cfg.n_note, cfg.n_velocity); vel[..., 2] = 1.0
        return (c, c + 1, c + 2, vel, None, c + 3, c + 4, c + 5, vel)

    outs = O.transcript(feat, fwd, cfg, min_value=-5.0)
    assert outs[0].shape == (24, 4)
    np.testing.assert_array_equal(outs[0][:n], feat)
    np.testing.assert_array_equal(outs[0][n:], np.full((3, 4), -5.0, np.float32))
    np.testing.assert_array_equal(outs[4][:n], feat + 3)
    assert outs[3].dtype == np.int8 and (outs[3] == 2).all()
    for n_off in (0, 2):
        outs = O.transcript_stride(feat, n_off, fwd, cfg, min_value=-5.0)
        assert outs[0].shape[1] == 4 and outs[0].shape[0] % 4 == 0
        np.testing.assert_array_equal(outs[1][:n], feat + 1)


def test_config5_paper_trained_golden():
    """Round 5: the CPU oracle against the REFERENCE's outputs on TRAINED paper-size weights (tests/golden/config5_paper_golden.npz, made by
    make_golden_r5.py from config5_paper_trained.npz: the paper-size model this repo's training step trained, two clips of config 5's minute).
    The six posterior tensors in full, samples of the velocity logits and of the attention tensor: 2e-6 of the tensor's scale."""
    import os, sys
    sys.path.insert(0, os.path.join(util.ROOT, 'tools'))
    from pack_checkpoint import unpack_state_dict
    g = util.golden('config5_paper_golden')
    sd = unpack_state_dict(np.load(os.path.join(util.GOLDEN, 'config5_paper_trained.npz')))
    assert len(sd) == 165 and sum(v.numel() for v in sd.values()) == 5516574
    torch.set_num_threads(min(8, torch.get_num_threads()))
    with torch.no_grad():
        out = O.model_forward(sd, torch.from_numpy(g['input']), O.PAPER)
    for n, t in zip(util.OUT_NAMES, out):
        if 'out.' + n in g.files:
            assert util.max_err(t, torch.from_numpy(g['out.' + n])) < 2e-6, n
        else:
            f = t.reshape(-1)
            ref = torch.from_numpy(g['out.' + n + '.sample'])
            assert util.max_err(f[::int(g['out.' + n + '.stride'])], ref) < 2e-6 * max(1.0, float(g['out.' + n + '.stats'][1])), n
    # the fixture's model means something: a trained transcriber is confident -- most posteriors sit near 0 or 1
    mpe = torch.from_numpy(g['out.mpe_B'])
    assert float(((mpe < 0.05) | (mpe > 0.95)).float().mean()) > 0.97 and 0.01 < float((mpe >= 0.5).float().mean()) < 0.2
