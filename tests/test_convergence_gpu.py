"""Training-quality parity (BASELINE.json metric: "... frame-F1 parity"): the same model trained from the same initial
parameters on the same clips in the device's precision modes -- and by the CPU oracle running the reference's own step
(training/train.py:89-160: forward, 6 x BCE + 2 x CE, backward, torch.optim.Adam) -- must follow the same loss trajectory and end at the
same frame-level F1 (mpe >= 0.5, evaluation/m_mpe.py:101, 166-175) with the same threshold-free ranking of the held-out frames (frame_auc).

The task is synthetic but LEARNABLE (the labels are a deterministic function of the spectrogram).  Round 5: the models start from the
reference's initialisation with the three POSITION tables multiplied by POS_SCALE = 300.  Why: m_training.py:31-33 draws nn.Embedding
tables xavier-uniform (|w| <= 0.11) and model_spec2midi.py:95,190 adds them to token embeddings x sqrt(hid_dim) whose entries are in the
hundreds on log-mel input, so as initialised the model cannot tell one bin (or frame) from another and sits on the predict-the-prior
plateau for thousands of steps -- in EVERY arithmetic, the fp32 CPU oracle included (measured there, mini size, 600 steps: held-out AUC
0.61 / F1 0.55 unscaled, 0.99 / 0.95 scaled; rounds 3-4 of this test trained on that plateau and could only assert AUC > 0.55).  With the
tables scaled the loss falls from 6.0 to below 2 within 600 steps and the held-out frame-F1 reaches 0.9+, so a mode whose gradients
pointed the wrong way fails by a wide margin: floors on F1 and AUC far above chance, the oracle comparison at the mini size
and at the PAPER'S WIDTH AND DEPTH (d 256, ff 512, 3+3 layers, 4 heads on short axes: VERDICT r04 item 2a)."""
import math

import numpy as np
import pytest
import torch

import util
from util import O, MINI

pytestmark = pytest.mark.gpu

# the paper's WIDTH (d = 256, ff = 512, 4 heads: the strip kernels of both modes) on short axes; DEEP adds the paper's depth (3 + 3 layers:
# DecoderLayer -- self(notes) + cross + FFN under one shared LayerNorm -- trained against the oracle at d = 256)
WIDE = O.HfttConfig(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, hid_dim=256, pf_dim=512, enc_layer=1, dec_layer=1,
                    enc_head=4, dec_head=4, n_note=8, n_velocity=16)
DEEP = O.HfttConfig(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, hid_dim=256, pf_dim=512, enc_layer=3, dec_layer=3,
                    enc_head=4, dec_head=4, n_note=8, n_velocity=16)
# the reference's DEFAULT width (m_training.py:56-61: d = 64, ff = 128): the small-width strip families of both modes (csrc/x3s_strip.h,
# csrc/bs_strip.hip) -- ADVICE r05: those kernels had kernel tests and a build comparison, but no learning evidence of their own
SMALL = O.HfttConfig(n_margin=4, n_frame=16, n_bin=48, cnn_channel=4, cnn_kernel=5, hid_dim=64, pf_dim=128, enc_layer=2, dec_layer=2,
                     enc_head=2, dec_head=2, n_note=12, n_velocity=16)
STEPS, EVERY = 600, 50          # (dropout-on legs: 2 x STEPS -- the masked model leaves the plateau about 300 steps later)
POS_SCALE = 300.0


def scaled_model(cfg, seed, dropout):
    model = util.build_model(cfg, seed, dropout=dropout)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'pos_embedding' in name:
                p.mul_(POS_SCALE)
    return model


def make_clips(cfg, n, seed):
    """log-mel-range spectrograms whose note activity is readable from the bins: note k is 'on' in frame t when the mean of its band of bins
    is above the clip-wide median; onset / offset targets are the rising / falling edges (soft), velocity the quantised band level."""
    g = torch.Generator().manual_seed(seed)
    W = cfg.n_frame + 2 * cfg.n_margin
    band = cfg.n_bin // cfg.n_note
    # slowly varying band envelopes + fine noise
    env = torch.randn(n, cfg.n_note, W // 4 + 2, generator=g)
    env = torch.nn.functional.interpolate(env, size=W, mode='linear', align_corners=True)
    spec = (env.repeat_interleave(band, dim=1) * 3.0 - 7.0 + 0.5 * torch.randn(n, cfg.n_note * band, W, generator=g))
    if spec.shape[1] < cfg.n_bin:
        spec = torch.cat([spec, torch.full((n, cfg.n_bin - spec.shape[1], W), -18.0)], 1)
    spec = spec.clamp(-18.420681, 6.0).contiguous()
    lvl = env[:, :, cfg.n_margin:cfg.n_margin + cfg.n_frame].transpose(1, 2)                      # [n, T, N]
    mpe = (lvl > 0.0).float()
    prev = torch.cat([mpe[:, :1], mpe[:, :-1]], 1)
    onset = ((mpe - prev) > 0).float()
    offset = ((prev - mpe) > 0).float()
    vel = ((lvl.clamp(-2, 2) + 2) / 4 * (cfg.n_velocity - 1)).round().long() * mpe.long()
    return spec, (onset.contiguous(), offset.contiguous(), mpe.contiguous(), vel.contiguous())


def frame_f1(prob, ref):
    est = prob >= 0.5
    ref = ref >= 0.5
    tp = float((est & ref).sum())
    prec = tp / max(float(est.sum()), 1.0)
    rec = tp / max(float(ref.sum()), 1.0)
    return 2 * prec * rec / max(prec + rec, 1e-12)


def frame_auc(prob, ref):
    """threshold-free: the fraction of (active, inactive) frame pairs the posterior ranks correctly (0.5 = chance).  The F1 at the fixed
    threshold 0.5 of a model this early in training swings with its calibration; the ranking does not."""
    p = prob.reshape(-1).double()
    y = ref.reshape(-1) >= 0.5
    order = torch.argsort(p)
    ranks = torch.empty_like(p)
    ranks[order] = torch.arange(1, p.numel() + 1, dtype=torch.float64)
    n1 = float(y.sum()); n0 = float(p.numel()) - n1
    if n1 == 0 or n0 == 0:
        return 0.5
    return float((ranks[y].sum() - n1 * (n1 + 1) / 2) / (n1 * n0))


def train_device(cfg, precision, dropout, data, held, dev, B, lr, STEPS=STEPS):
    from hftt_hip.trainer import TrainStep
    model = scaled_model(cfg, 2025, dropout).to(dev)
    model.hftt_precision = precision
    model.train()
    ts = TrainStep(model, lr=lr)
    spec, labels = data
    n = spec.shape[0]
    curve = []
    acc = 0.0
    f1s = []
    for s in range(STEPS):
        idx = [(s * B + i) % n for i in range(B)]
        loss = ts(spec[idx].to(dev), *[t[idx].to(dev).contiguous() for t in labels])
        if s == 0:                                    # the reference's default width runs on the small-width strip families in both modes
            assert ts.engine.strip_small == (cfg.hid_dim == 64 and cfg.pf_dim == 128 and precision in ('x3', 'bf16'))
        acc += float(loss[0])
        if (s + 1) % EVERY == 0:
            curve.append(acc / EVERY); acc = 0.0
            print('  %s step %d loss %.4f' % (precision, s + 1, curve[-1]), flush=True)
            if s + 1 > STEPS - 3 * EVERY:             # held-out frame-F1 at the last three checkpoints (the dropout counter only moves in training mode)
                model.eval()
                with torch.no_grad():
                    out = model(held[0].to(dev))
                f1s.append((frame_f1(out[7].cpu(), held[1][2]), frame_f1(out[2].cpu(), held[1][2]), frame_auc(out[7].cpu(), held[1][2])))
                model.train()
    # the median over the last three checkpoints is what the comparisons below use
    med = lambda v: sorted(v)[len(v) // 2]
    print('  %s held-out F1 (B, A) and ranking AUC (B) at the last checkpoints: %s' % (precision, [tuple(round(v, 3) for v in t) for t in f1s]), flush=True)
    return curve, med([t[0] for t in f1s]), med([t[1] for t in f1s]), model, med([t[2] for t in f1s])


def train_oracle(cfg, data, held, B, lr):
    """the reference's step on the CPU: oracle forward (dropout 0), train.py:141-153 loss, autograd, torch.optim.Adam(lr)"""
    torch.set_num_threads(min(8, torch.get_num_threads()))        # small tensors: more threads only add overhead
    model = scaled_model(cfg, 2025, 0.0)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=lr)
    spec, labels = data
    n = spec.shape[0]
    curve, acc, f1s = [], 0.0, []
    for s in range(STEPS):
        idx = [(s * B + i) % n for i in range(B)]
        opt.zero_grad()
        loss = O.spec2midi_loss(O.model_forward(sd, spec[idx], cfg), *[t[idx] for t in labels])
        loss.backward()
        opt.step()
        acc += float(loss.detach())
        if (s + 1) % EVERY == 0:
            curve.append(acc / EVERY); acc = 0.0
            print('  oracle step %d loss %.4f' % (s + 1, curve[-1]), flush=True)
            if s + 1 > STEPS - 3 * EVERY:
                with torch.no_grad():
                    out = O.model_forward(sd, held[0], cfg)
                f1s.append((frame_f1(out[7], held[1][2]), frame_f1(out[2], held[1][2]), frame_auc(out[7], held[1][2])))
    med = lambda v: sorted(v)[len(v) // 2]
    return curve, med([t[0] for t in f1s]), med([t[1] for t in f1s]), None, med([t[2] for t in f1s])


def first_below(curve, level):
    """index of the first checkpoint whose mean loss is below `level` (len(curve) if none): WHEN the run left the plateau"""
    for i, v in enumerate(curve):
        if v < level:
            return i
    return len(curve)


@pytest.mark.parametrize('dropout', [0.0, 0.1])
@pytest.mark.parametrize('size', ['mini', 'small', 'wide', 'deep'])
def test_modes_train_alike(dev, size, dropout):
    cfg = {'mini': MINI, 'small': SMALL, 'wide': WIDE, 'deep': DEEP}[size]
    B = 4
    lr = 1e-3 if size in ('mini', 'small') else 3e-4          # (the 256-wide model diverges at 1e-3 in every mode, the oracle's fp32 included)
    data = make_clips(cfg, 64, seed=1)
    held = make_clips(cfg, 48, seed=2)
    res = {m: train_device(cfg, m, dropout, data, held, dev, B, lr, STEPS if dropout == 0.0 else 2 * STEPS) for m in ('x3', 'bf16')}
    if size in ('mini', 'small', 'deep') and dropout == 0.0:
        res['oracle'] = train_oracle(cfg, data, held, B, lr)
    rep = {m: {'loss': [round(v, 4) for v in r[0]], 'f1_B': round(r[1], 4), 'f1_A': round(r[2], 4), 'auc_B': round(r[4], 4)} for m, r in res.items()}
    print(size, 'dropout', dropout, rep)
    for m, r in res.items():
        assert all(math.isfinite(v) for v in r[0]), (m, r[0])
        # every mode LEARNS the task: the loss leaves the plateau (6.0 -> below 3.5; the floor is the entropy of the velocity classes) and the
        # held-out frame decisions are right -- floors far above chance (F1 of the all-on answer: 0.67, AUC 0.5)
        assert r[0][-1] < 3.5 and r[0][-1] < 0.6 * r[0][0], (m, r[0])
        assert r[1] > 0.85 and r[2] > 0.85, 'held-out frame-F1 (B, A) of %s: %s' % (m, r[1:3])
        assert r[4] > 0.97, 'held-out ranking AUC of %s: %g' % (m, r[4])
    base = res['x3']
    for m, r in res.items():
        if m == 'x3':
            continue
        # Same trajectory: on the plateau the curves agree to a per cent; the step at which a run LEAVES it is the sensitive quantity (the
        # loss then falls by a third within 100 steps, so a shift of a few steps is a large difference at one checkpoint): it may move by one
        # checkpoint (50 steps) against x3, and the end points -- loss, F1, ranking -- must agree.
        lvl = 0.5 * (base[0][0] + base[0][-1])
        ia, ib = first_below(r[0], lvl), first_below(base[0], lvl)
        if m == 'oracle':
            # measured (round 5): the same checkpoint at the mini size and at the paper's width + depth; before it 0.2 %, end loss 10 - 18 %
            assert abs(ia - ib) <= 1, (m, ia, ib, r[0], base[0])
            for i in range(min(ia, ib) - 2):
                assert abs(r[0][i] - base[0][i]) <= 0.015 * base[0][i], (m, i, r[0], base[0])
            assert abs(r[0][-1] - base[0][-1]) <= 0.25 * base[0][-1], (m, r[0][-1], base[0][-1])
        else:
            # The single-pass bf16 mode is the throughput mode and claims no output parity, but it must TRAIN: the floors above hold for it.
            # WHEN it leaves the plateau is looser than for the fp32-class modes: without dropout at the same checkpoint as x3 (+- 1), with
            # dropout up to 5 checkpoints (250 steps) later at the paper's depth -- its gradient carries more rounding noise (DESIGN section 2,
            # round 5).
            assert abs(ia - ib) <= (2 if dropout == 0.0 else 6), (m, ia, ib, r[0], base[0])
            for i in range(min(ia, ib) - 2):
                assert abs(r[0][i] - base[0][i]) <= 0.03 * base[0][i], (m, i, r[0], base[0])
            # end loss at EQUAL TIME SINCE LEAVING THE PLATEAU (a run that left it k checkpoints later is compared with x3 k checkpoints before
            # its end): within a third
            lag = max(0, ia - ib)
            ref_end = base[0][len(base[0]) - 1 - lag]
            assert abs(r[0][-1] - ref_end) <= 0.35 * ref_end, (m, r[0][-1], ref_end, lag)
        assert abs(r[1] - base[1]) <= 0.06 and abs(r[2] - base[2]) <= 0.06, (m, r[1:3], base[1:3])
        assert abs(r[4] - base[4]) <= 0.02, (m, r[4], base[4])


def test_plane_and_fp32_operand_backward_agree_on_the_training_shapes(dev, monkeypatch):
    """the f16-pair planes changed the x3 backward's operands (bf16 pairs rebuilt from hi + lo instead of split from the fp32
    value).  On the training shapes of the wide model, after 100 steps of training (not at initialisation), every gradient tensor of the
    plane plan agrees with the fp32-operand plan's to 2e-3 of the tensor's maximum (measured: 1.1e-3 on the first-layer tensors, 7e-4 behind
    them; kernel-level: tests/test_x3_gpu.py, 4e-5 -- the model amplifies the attention kernels' last-bit differences through the first
    layer's near-one-hot softmax, as it does between any two fp32-class arithmetics: the oracle comparison allows those tensors 2e-2)."""
    from hftt_hip.trainer import TrainStep
    cfg, B = WIDE, 4
    spec, labels = make_clips(cfg, 64, seed=1)
    model = scaled_model(cfg, 2025, 0.1).to(dev)
    model.hftt_precision = 'x3'
    model.train()
    ts = TrainStep(model, lr=3e-4)
    for s_ in range(100):
        idx = [(s_ * B + i) % 64 for i in range(B)]
        ts(spec[idx].to(dev), *[t[idx].to(dev).contiguous() for t in labels])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {}
    idx = list(range(B))
    for planes in ('1', '0'):
        monkeypatch.setenv('HFTT_X3_PLANES', planes)
        m2 = util.build_model(cfg, 1, dropout=0.1)
        m2.load_state_dict(sd)
        m2 = m2.to(dev)
        m2.hftt_precision = 'x3'
        m2.train()
        t2 = TrainStep(m2, lr=3e-4)
        assert t2.engine.planes_opt == (planes == '1')
        t2.engine.step_counter = 1000                 # the same dropout masks in both plans
        t2.forward_backward(spec[idx].to(dev), *[t[idx].to(dev).contiguous() for t in labels])
        grads[planes] = {n: t2.engine.flat_grads[o:o + k].clone() for (n, _p, o, k) in t2.engine._bound}
    # The tensors behind the FIRST encoder layer's attention are ill-conditioned in any arithmetic (its logits reach 1e4 .. 1e5 on log-mel
    # input x sqrt(d): a last-bit change of an operand re-weights near-tied keys -- DESIGN section 3, round-2 findings): their bound is the
    # one the oracle comparison uses for them; everything downstream of a soft attention agrees to the pair's precision.
    first = ('conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq', 'layers_freq.0.self_attention.fc_q', 'layers_freq.0.self_attention.fc_k')
    worst = {True: (0.0, ''), False: (0.0, '')}
    for n, a in grads['1'].items():
        b = grads['0'][n]
        scale = float(b.abs().max())
        if scale < 1e-9 or n.endswith('fc_k.bias'):           # (a key bias has a zero true gradient: what is computed is rounding residue)
            continue
        cls = n.startswith('encoder') and any(t in n for t in first)
        e = float((a - b).abs().max()) / scale
        if e > worst[cls][0]:
            worst[cls] = (e, n)
    print('  plane vs fp32-operand plan, worst gradient difference / tensor maximum: first-layer tensors %.2e (%s), others %.2e (%s)'
          % (worst[True] + worst[False]))
    assert worst[True][0] <= 3e-3 and worst[False][0] <= 2e-3, worst


def test_paper_size_modes_train_alike(dev):
    """BASELINE config 3 itself (paper size, batch 8, dropout 0.1, Adam lr 1e-4 as m_training.py's default): 200 steps from the same initial
    parameters on the same clips in the exact-fp32 mode, the benchmarked x3 mode and the bf16 mode -- the loss averaged over each 50 steps.
    Measured (8.557 -> 7.740 in the exact mode): x3 within 0.03 % of it at every checkpoint, bf16 within 0.25 %."""
    from hftt_hip.trainer import TrainStep
    cfg = O.PAPER
    B, steps, every = 8, 200, 50
    data = make_clips(cfg, 32, seed=3)
    spec, labels = data
    curves = {}
    for mode in ('parity', 'x3', 'bf16'):
        model = util.build_model(cfg, 2025, dropout=0.1).to(dev)
        model.hftt_precision = mode
        model.train()
        ts = TrainStep(model, lr=1e-4)
        curve, acc = [], 0.0
        for s_ in range(steps):
            idx = [(s_ * B + i) % spec.shape[0] for i in range(B)]
            loss = ts(spec[idx].to(dev), *[t[idx].to(dev).contiguous() for t in labels])
            acc += float(loss[0])
            if (s_ + 1) % every == 0:
                curve.append(acc / every); acc = 0.0
        curves[mode] = curve
        print('  paper size', mode, [round(v, 4) for v in curve], flush=True)
        ts.engine._ws.clear()
        del ts, model
        torch.cuda.empty_cache()
    base = curves['parity']
    assert base[-1] < base[0] - 0.3, base
    for a, b in zip(curves['x3'], base):
        assert abs(a - b) <= 0.003 * b, (curves['x3'], base)
    for a, b in zip(curves['bf16'], base):
        assert abs(a - b) <= 0.02 * b, (curves['bf16'], base)
