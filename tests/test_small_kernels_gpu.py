"""The small kernels of the path on their own, through the C ABI (ctypes), against fp64 torch restatements: loss (BCE with saturated
probabilities, a velocity vocabulary other than 128), heads split / backward in both row orders, the time-axis transpose + positional
embedding + dropout and its backward, the window gather, the embedding fold and its backward, the in-place dropout backward.
(Until round 2 these ran only inside the whole-model MINI test at d = 64.)"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util
from util import O, keep_scale, rel_err, max_err, keep_mask_t

pytestmark = pytest.mark.gpu


def _lib():
    from hftt_hip import _capi
    return _capi, _capi.lib()


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


@pytest.mark.parametrize('V,n', [(128, 8 * 16 * 11), (128, 1001), (16, 1000), (12, 77), (5, 37), (130, 50)])
def test_loss_kernel_values_and_gradients(dev, V, n):
    """training/train.py:141-153: 6x BCELoss(mean) (log clamped at -100, like torch) + 2x CrossEntropyLoss(mean), weights wA / wB."""
    capi, L = _lib()
    g = torch.Generator().manual_seed(V)
    probs = [torch.rand(n, generator=g) for _ in range(6)]
    for p in probs:                                     # saturated posteriors: exactly 0 and exactly 1, against both kinds of target
        p[:4] = torch.tensor([0.0, 1.0, 0.0, 1.0])
    vel = [torch.randn(n, V, generator=g) * 3 for _ in range(2)]
    lo, lf = torch.rand(n, generator=g), torch.rand(n, generator=g)
    lo[:4] = torch.tensor([0.0, 1.0, 1.0, 0.0]); lf[:4] = torch.tensor([1.0, 0.0, 0.0, 1.0])
    lm = (torch.rand(n, generator=g) < 0.3).float()
    lv = torch.randint(0, V, (n,), generator=g)
    wA, wB = 0.7, 1.3
    p64 = [p.double().requires_grad_(True) for p in probs]
    v64 = [v.double().requires_grad_(True) for v in vel]
    terms = [F.binary_cross_entropy(p64[0], lo.double()), F.binary_cross_entropy(p64[1], lf.double()), F.binary_cross_entropy(p64[2], lm.double()),
             F.cross_entropy(v64[0], lv),
             F.binary_cross_entropy(p64[3], lo.double()), F.binary_cross_entropy(p64[4], lf.double()), F.binary_cross_entropy(p64[5], lm.double()),
             F.cross_entropy(v64[1], lv)]
    total = wA * sum(terms[:4]) + wB * sum(terms[4:])
    total.backward()
    d = capi.LossDesc()
    d.n, d.V = n, V
    dp = [p.to(dev).contiguous() for p in probs]; dv = [v.to(dev).contiguous() for v in vel]
    gp = [torch.full((n,), float('nan'), device=dev) for _ in range(6)]; gv = [torch.full((n, V), float('nan'), device=dev) for _ in range(2)]
    for i in range(6):
        d.prob[i], d.d_prob[i] = dp[i].data_ptr(), gp[i].data_ptr()
    for i in range(2):
        d.vel[i], d.d_vel[i] = dv[i].data_ptr(), gv[i].data_ptr()
    labs = (lo.to(dev), lf.to(dev), lm.to(dev), lv.to(dev))
    d.label_onset, d.label_offset, d.label_mpe, d.label_velocity = (t.data_ptr() for t in labs)
    d.weight_A, d.weight_B, d.grad_scale = wA, wB, 1.0
    out = torch.zeros(16, device=dev)
    ws = torch.empty(L.hftt_loss_ws_bytes(n) // 4 + 16, device=dev)
    d.loss_out, d.ws = out.data_ptr(), ws.data_ptr()
    capi.check(L.hftt_loss(C.byref(d), _st(dev)), 'loss')
    got = out[:9].cpu().double()
    assert abs(got[0] - total.item()) < 2e-5 * abs(total.item())
    order = [0, 1, 2, 4, 5, 6, 3, 7]                        # loss_out[1..8]: the six BCE terms in prob[] order, then ... (checked below as a set)
    ref_terms = sorted(t.item() for t in terms)
    assert np.allclose(sorted(got[1:9].tolist()), ref_terms, rtol=2e-5, atol=1e-6)
    # gradients; a saturated posterior on the wrong side has the clamped-log gradient torch gives (huge but finite): compare relative
    for i, k in enumerate((0, 1, 2, 3, 4, 5)):
        ref = p64[k].grad
        assert torch.isfinite(gp[i]).all()
        ok = (gp[i].cpu().double() - ref).abs() <= 2e-5 * ref.abs() + 1e-9
        assert ok.all(), (i, int((~ok).sum()))
    for i in range(2):
        assert rel_err(gv[i], v64[i].grad) < 2e-5


@pytest.mark.parametrize('tm', [0, 1])
@pytest.mark.parametrize('V', [128, 16])
def test_heads_split_and_backward(dev, tm, V):
    """model_spec2midi.py:172-175 (rows (b,t,n)) and :203-206 (rows (b,n,t), outputs permuted back to (b,t,n))."""
    capi, L = _lib()
    B, T, N = 3, 16, 11
    ldl = ((V + 3 + 63) // 64) * 64
    g = torch.Generator().manual_seed(V + tm)
    S = B * T * N
    logits = torch.randn(S, ldl, generator=g) * 2
    x64 = logits.double().requires_grad_(True)

    def ref(x):
        r = x.view(B, N, T, ldl).permute(0, 2, 1, 3) if tm else x.view(B, T, N, ldl)
        return torch.sigmoid(r[..., V]), torch.sigmoid(r[..., V + 1]), torch.sigmoid(r[..., V + 2]), r[..., :V]
    on, of, mp, ve = ref(x64)
    dl = logits.to(dev)
    o = [torch.empty(B, T, N, device=dev) for _ in range(3)] + [torch.empty(B, T, N, V, device=dev)]
    capi.check(L.hftt_heads_split(dl.data_ptr(), ldl, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), B, T, N, V, tm, _st(dev)), 'heads')
    for a, b in zip(o, (on, of, mp, ve)):
        assert max_err(a, b) < 2e-6
    grads = [torch.randn(B, T, N, generator=g) for _ in range(3)] + [torch.randn(B, T, N, V, generator=g)]
    (on * grads[0].double()).sum().backward(retain_graph=True)
    (of * grads[1].double() + mp * grads[2].double()).sum().backward(retain_graph=True)
    (ve * grads[3].double()).sum().backward()
    dg = [t.to(dev).contiguous() for t in grads]
    dlog = torch.full((S, ldl), float('nan'), device=dev)
    capi.check(L.hftt_heads_split_bwd(o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(), dg[2].data_ptr(),
                                      dg[3].data_ptr(), dlog.data_ptr(), ldl, B, T, N, V, tm, _st(dev)), 'heads_bwd')
    assert max_err(dlog[:, :V + 3], x64.grad[:, :V + 3]) < 2e-6
    assert torch.isfinite(dlog).all() and float(dlog[:, V + 3:].abs().max()) == 0.0          # padding columns get zeros (they feed a GEMM)


@pytest.mark.parametrize('p', [0.0, 0.25])
@pytest.mark.parametrize('half', [False, True])
def test_time_embed_forward_backward(dev, p, half):
    """model_spec2midi.py:189-191: [B,T,N,d] -> [B*N,T,d], * sqrt(d) + pos_embedding_time[t], dropout; and its backward."""
    capi, L = _lib()
    B, T, N, d = 2, 16, 11, 64
    g = torch.Generator().manual_seed(int(p * 100) + half)
    dt = torch.bfloat16 if half else torch.float32
    x = torch.randn(B * T, N, d, generator=g).to(dt)
    pos = torch.randn(T, d, generator=g)
    site, seed, scale = 9, 4321, math.sqrt(d)
    fl = (capi.TE_X_BF16 | capi.TE_Y_BF16) if half else 0
    y = torch.empty(B * N, T, d, device=dev, dtype=dt)
    xd, posd = x.to(dev), pos.to(dev)                  # (named: a temporary's memory is recycled before the kernel runs)
    capi.check(L.hftt_time_embed_fwd(xd.data_ptr(), posd.data_ptr(), y.data_ptr(), B, T, N, d, scale, p, site, seed, fl, _st(dev)), 'te')
    ref = x.double().view(B, T, N, d).permute(0, 2, 1, 3).reshape(B * N, T, d) * scale + pos.double()[None]
    mask = keep_mask_t(seed, site, (B * N, T, d), p).double() if p > 0 else torch.ones(B * N, T, d, dtype=torch.float64)
    ref = ref * mask * keep_scale(p)
    assert rel_err(y, ref) < (6e-3 if half else 2e-6)
    # backward: dx[(b,t),n,:] = mask * dy[(b,n),t,:] * scale / (1-p); dym = masked dy (for the positional table's column sum)
    dy = torch.randn(B * N, T, d, generator=g).to(dt)
    flb = (capi.TE_X_BF16 | capi.TE_Y_BF16 | capi.TE_M_BF16) if half else 0
    dx = torch.empty(B * T, N, d, device=dev, dtype=dt)
    dym = torch.empty(B * N, T, d, device=dev, dtype=dt)
    dyd = dy.to(dev)
    capi.check(L.hftt_time_embed_bwd(dyd.data_ptr(), dx.data_ptr(), dym.data_ptr(), B, T, N, d, scale, p, site, seed, 0, flb, _st(dev)), 'te_bwd')
    gm = dy.double() * mask * keep_scale(p)
    assert rel_err(dym, gm) < (6e-3 if half else 2e-6)
    assert rel_err(dx, (gm * scale).view(B, N, T, d).permute(0, 2, 1, 3).reshape(B * T, N, d)) < (6e-3 if half else 2e-6)


def test_im2win_and_embedding_fold(dev):
    """Encoder front (model_spec2midi.py:65-85): unfold + Conv2d(1,C,(1,k)) + flatten + Linear == window gather x folded weights; and the fold's
    backward maps (dWeff, dbeff) onto the four reference parameters exactly as autograd does."""
    capi, L = _lib()
    B, Fq, T, M, Cc, kw, d = 2, 12, 16, 4, 4, 5, 48
    n_proc = 2 * M + 1
    nw = n_proc - kw + 1
    Kp = ((n_proc + 31) // 32) * 32
    d_pad = ((d + 63) // 64) * 64
    g = torch.Generator().manual_seed(3)
    spec = torch.randn(B, Fq, T + 2 * M, generator=g)
    wconv = torch.randn(Cc, 1, 1, kw, generator=g).double().requires_grad_(True)
    bconv = torch.randn(Cc, generator=g).double().requires_grad_(True)
    wtok = (torch.randn(d, Cc * nw, generator=g) / 4).double().requires_grad_(True)
    btok = torch.randn(d, generator=g).double().requires_grad_(True)
    # reference arithmetic of the encoder front, fp64
    win = spec.double().unfold(2, n_proc, 1).permute(0, 2, 1, 3).contiguous()                      # [B, T, F, n_proc]
    conv = F.conv2d(win.reshape(B * T, 1, Fq, n_proc), wconv, bconv)                                # [B*T, C, F, nw]
    tok = F.linear(conv.permute(0, 2, 1, 3).reshape(B * T, Fq, Cc * nw), wtok, btok)                # [B*T, F, d]
    # the window gather
    A = torch.full((B * T * Fq, Kp), float('nan'), device=dev)
    specd = spec.to(dev)
    capi.check(L.hftt_im2win(specd.data_ptr(), A.data_ptr(), B, Fq, T, n_proc, Kp, _st(dev)), 'im2win')
    assert torch.equal(A[:, :n_proc].cpu(), win.reshape(B * T * Fq, n_proc).float()) and float(A[:, n_proc:].abs().max()) == 0.0
    # the fold
    f = capi.FoldDesc()
    f.d, f.C, f.kw, f.n_proc, f.Kp, f.d_pad = d, Cc, kw, n_proc, Kp, d_pad
    dev_t = [t.detach().float().to(dev).contiguous() for t in (wconv.reshape(Cc, kw), bconv, wtok, btok)]
    f.wconv, f.bconv, f.wtok, f.btok = (t.data_ptr() for t in dev_t)
    weff = torch.full((d_pad, Kp), float('nan'), device=dev)
    beff = torch.empty(d, device=dev)
    f.weff_bf, f.weff_f32, f.beff = None, weff.data_ptr(), beff.data_ptr()
    capi.check(L.hftt_embed_fold_fwd(C.byref(f), _st(dev)), 'fold')
    got = A.double().cpu() @ weff[:d].double().cpu().T + beff.double().cpu()
    assert rel_err(got, tok.reshape(B * T * Fq, d)) < 1e-5
    assert float(weff[:d, n_proc:].abs().max()) == 0.0 and (d == d_pad or float(weff[d:].abs().max()) == 0.0)      # zero padding feeds the GEMM
    # backward: feed dWeff = A^T dTok, dbeff = colsum(dTok) and compare the four parameter gradients with autograd's
    dtok = torch.randn(B * T * Fq, d, generator=g).double()
    (tok.reshape(B * T * Fq, d) * dtok).sum().backward()
    dweff = (dtok.T @ A.double().cpu()).float().to(dev).contiguous()                                # [d, Kp]
    dbeff = dtok.sum(0).float().to(dev)
    gr = [torch.full_like(t, float('nan')) for t in dev_t]
    f.dweff, f.dbeff = dweff.data_ptr(), dbeff.data_ptr()
    f.g_wconv, f.g_bconv, f.g_wtok, f.g_btok = (t.data_ptr() for t in gr)
    capi.check(L.hftt_embed_fold_bwd(C.byref(f), _st(dev)), 'fold_bwd')
    for a, b in zip(gr, (wconv.grad.reshape(Cc, kw), bconv.grad, wtok.grad, btok.grad)):
        assert rel_err(a, b) < 2e-5


@pytest.mark.parametrize('half', [False, True])
def test_dropout_bwd_in_place(dev, half):
    capi, L = _lib()
    M, N, p, site, seed = 333, 256, 0.1, 17, 99
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, N, generator=g).to(torch.bfloat16 if half else torch.float32)
    buf = x.to(dev).clone()
    capi.check(L.hftt_dropout_bwd(buf.data_ptr(), M * N, p, site, seed, 1 if half else 0, _st(dev)), 'dropout_bwd')
    mask = keep_mask_t(seed, site, (M, N), p)
    ref = x.double() * mask.double() * keep_scale(p)
    assert rel_err(buf, ref) < (6e-3 if half else 1e-6)
    assert torch.equal((buf == 0).cpu() | mask, torch.ones(M, N, dtype=torch.bool))
    # the quantised keep rate: thr = round(0.9 * 256) = 230 of 256 (csrc/hftt_common.h), within sampling noise
    rate = mask.double().mean().item()
    assert abs(rate - 230 / 256) < 4 * math.sqrt(0.9 * 0.1 / (M * N))
    with pytest.raises(capi.HfttError):
        capi.check(L.hftt_dropout_bwd(buf.data_ptr(), M * N + 2, p, site, seed, 0, _st(dev)), 'dropout_bwd')


@pytest.mark.parametrize('sr_in', [44100, 48000, 22050, 8000, 32000])
def test_resample_kernel_against_the_float64_restatement(dev, sr_in):
    """hftt_resample (model/amt.py:57-58: Resample(sr, 16000)) against oracle.resample on the same samples: output length = ceil(n * 16000 / sr),
    every sample within 2e-6 of the float64 polyphase sum (fp32 accumulation over <= 475 taps of a unit-gain filter); odd lengths, the zero
    padding at both borders included.  Then through AMT.wave2feature: a 44.1 kHz wave gives the features of its resampled self."""
    from hftt_hip import ops
    g = torch.Generator().manual_seed(sr_in)
    for n in (sr_in // 3 + 7, 1001):
        x = (torch.rand(n, generator=g) * 2 - 1).float()
        y = ops.resample(x.to(dev), sr_in, 16000).cpu()
        ref = O.resample(x, sr_in, 16000)
        assert y.shape == ref.shape and y.numel() == -(-n * 16000 // sr_in)
        assert max_err(y, ref) < 2e-6 * 4, (sr_in, n, max_err(y, ref))


def test_wave2feature_resamples_on_the_device(dev, tmp_path):
    import pickle
    from model.amt import AMT
    from corpus import synth_audio as SA
    f = tmp_path / 'm.pkl'
    with open(f, 'wb') as fh:
        pickle.dump(util.build_model(O.MICRO, 1), fh, protocol=4)
    amt = AMT(SA.default_config(), str(f))
    t = torch.arange(44100, dtype=torch.float64) / 44100.0
    wave = (0.3 * torch.sin(2 * math.pi * 440.0 * t) + 0.2 * torch.sin(2 * math.pi * 1234.5 * t)
            + 0.01 * torch.randn(44100, generator=torch.Generator().manual_seed(9), dtype=torch.float64)).float()    # (a noise floor as in test_logmel:
    # in the empty bins of a pure two-tone signal the log amplifies the last bit of the samples)
    feat = amt.wave2feature(wave, 44100)
    ref = O.logmel(O.resample(wave, 44100, 16000))
    assert feat.shape == ref.shape == (1 + 16000 // 256, 256)
    assert max_err(feat, ref) < 5e-3          # log of a power spectrum: the fp32 FFT's relative error (test_logmel's bound)
