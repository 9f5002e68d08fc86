"""The BENCHMARKED arithmetic (hftt_precision = 'bf16', the mode bench.py times) at the paper size, against
  (a) the reference-generated golden fixture tests/golden/paper_b1.npz (same seed / inputs as test_golden_fixture),
  (b) this repo's <= 1e-3 parity mode on the device at B = 8 (BASELINE config 3: full tensors, loss, every gradient),
  (c) BASELINE config 5 at full size: 60 s of audio = 30 clips through model.amt.AMT (decode determinism, outputs vs parity mode).
Budgets: SURVEY.md section 7 measured, on the reference itself, what bf16 arithmetic costs at this size -- bf16-rounded GEMM operands
with fp32 accumulate: posteriors 2.5e-2, velocity logits 1.8e-1, attention 6.8e-3; torch.autocast (bf16 GEMMs AND a bf16 residual
stream, which is what this build's bf16 mode stores): 2.8e-2 / 2.0e-1 / 6.8e-3.  Those were one seed; the budgets below are the
autocast row with 1.4x headroom for other seeds / inputs, and every test prints what it measured."""
import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch

import util
from util import O, OUT_NAMES, max_err

pytestmark = pytest.mark.gpu

LOSS_REL = 5e-3          # the velocity cross-entropies (ln 128 each at init) carry the logit error: relative, not absolute
BUDGET = {'onset_A': 5e-2, 'offset_A': 5e-2, 'mpe_A': 5e-2, 'onset_B': 5e-2, 'offset_B': 5e-2, 'mpe_B': 5e-2,      # (measured 2.9e-2 .. 4.1e-2)
          'velocity_A': 2.8e-1, 'velocity_B': 2.8e-1, 'attention': 2.5e-2}


def _to_dev(labels, dev):
    return tuple(t.to(dev).contiguous() for t in labels)


def _cos(a, b):
    a = a.double().reshape(-1); b = b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def test_bf16_mode_against_the_paper_size_golden(dev):
    """hftt_precision='bf16' vs reference outputs / loss / gradient samples of golden/paper_b1 (no oracle in the loop)."""
    from hftt_hip.trainer import TrainStep
    g = util.golden('paper_b1')
    cfg = util.cfg_from_golden(g)
    seed, B = int(g['seed']), int(g['bsz'])
    model = util.build_model(cfg, seed)
    util.perturb(model, seed + 1)
    model = model.to(dev)
    model.hftt_precision = 'bf16'
    model.train()                                 # dropout 0: identical to eval
    x = O.synth_spec(B, cfg, salt=seed)
    labels = O.synth_labels(B, cfg, salt=seed + 7)
    ts = TrainStep(model)
    ts.engine.flat_grads.fill_(float('nan'))
    loss = ts.forward_backward(x.to(dev), *_to_dev(labels, dev))
    outs = ts.engine._ws[B]['outs']
    rep = {}
    for n, t in zip(OUT_NAMES, outs):
        st = int(g['out.' + n + '.stride'])
        ref = torch.from_numpy(g['out.' + n + '.sample'])
        mine = t.reshape(-1)[::st].cpu()
        rep[n] = max_err(mine, ref)
        if n.startswith('mpe'):
            # thresholded frame decisions (evaluation/m_mpe.py:101) may differ from the reference only inside the error band
            flip = (mine >= 0.5) != (ref >= 0.5)
            band = (ref[flip] - 0.5).abs().max().item() if flip.any() else 0.0
            rep[n + '.flipped'] = float(flip.float().mean())
            assert band <= rep[n] + 1e-6
    rep['loss_rel_err'] = abs(loss[0].item() - float(g['loss'])) / float(g['loss'])
    cos = {}
    for (pname, _, o, n) in ts.engine._bound:
        gr = ts.engine.flat_grads[o:o + n].cpu()
        assert torch.isfinite(gr).all(), pname
        stats = g['grad.' + pname + '.stats']
        if stats[1] < 1e-7:                       # fc_k.bias: structurally zero gradient
            continue
        st = max(1, n // 64) | 1
        c = _cos(gr[::st], torch.from_numpy(g['grad.' + pname + '.sample']))
        # first-layer tensors are ill-conditioned on raw log-mel input (see test_golden_fixture); everything else is tight
        first = pname.startswith('encoder') and any(t in pname for t in ('conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq',
                                                                         'layers_freq.0.self_attention.fc_q', 'layers_freq.0.self_attention.fc_k'))
        cos[pname] = (c, first, abs(gr.double().norm().item() - stats[2]) / stats[2])
    _report_cosines(rep, {n: (c, f) for n, (c, f, _) in cos.items()})
    print('bf16 mode vs paper_b1 golden:', json.dumps(rep))
    for n in OUT_NAMES:
        assert rep[n] <= BUDGET[n], (n, rep[n])
    assert rep['loss_rel_err'] < LOSS_REL
    _assert_cosines(rep, full=False)


def _report_cosines(rep, cos):
    """cos: name -> (cosine vs the fp32 gradient, is a first-layer tensor)"""
    down = sorted(c for c, f in cos.values() if not f)
    rep['grad_cos.first_layer_tensors'] = sorted((round(c, 4), n) for n, (c, f) in cos.items() if f)
    rep['grad_cos.downstream.worst'] = sorted((round(c, 4), n) for n, (c, f) in cos.items() if not f)[:6]
    rep['grad_cos.downstream.count'] = len(down)
    for thr in (0.999, 0.99, 0.9):
        rep['grad_cos.downstream.frac>=%g' % thr] = round(sum(c >= thr for c in down) / len(down), 4)
    rep['grad_cos.downstream.median'] = round(down[len(down) // 2], 5)


def _assert_cosines(rep, full):
    # What bf16 arithmetic can and cannot track at random init (documented in DESIGN.md section 2):
    #  * tensors fed THROUGH the first encoder layer's attention (conv, tok_embedding, pos_embedding, layer-0 fc_q/fc_k): on raw log-mel
    #    input its logits reach ~1e4, the softmax is one-hot except at near-ties, and those near-tie terms -- which dominate these
    #    gradients -- are re-ranked by a bf16 product: cosine vs fp32 is NOT near 1 (printed, not asserted; the parity mode is exact);
    #  * fc_q / fc_k of an attention whose keys are near-identical across the sequence (decoder self-attention over the 88 note
    #    queries of one frame): dW = sum_k dk_k (x) x_k with sum_k dk_k = 0 is a difference of large cancelling terms.
    # Measured (round 2, B = 8, full tensors): median cosine 0.999, 88 % of the 146 downstream tensors >= 0.99, 96 % >= 0.9; the
    # fixture variant sees 64 strided samples per tensor at B = 1 and is correspondingly noisier.  The thresholds are regression guards
    # for that measured state, not a claim that bf16 operands reproduce every fp32 gradient.
    if full:
        assert rep['grad_cos.downstream.frac>=0.99'] >= 0.80 and rep['grad_cos.downstream.frac>=0.9'] >= 0.93, rep['grad_cos.downstream.worst']
        assert rep['grad_cos.downstream.median'] >= 0.998
    else:
        assert rep['grad_cos.downstream.frac>=0.9'] >= 0.90, rep['grad_cos.downstream.worst']
        assert rep['grad_cos.downstream.median'] >= 0.95


def test_bf16_mode_paper_b8_against_parity_mode(dev):
    """BASELINE config 3 (paper size, batch 8): full tensors of the benchmarked mode vs the <= 1e-3 mode on the same device."""
    from hftt_hip.trainer import TrainStep
    cfg, B = O.PAPER, 8
    model = util.build_model(cfg, 2024)
    util.perturb(model, 2025)
    model = model.to(dev)
    model.train()
    x = O.synth_spec(B, cfg, salt=31).to(dev)
    labels = _to_dev(O.synth_labels(B, cfg, salt=32), dev)
    res = {}
    for mode in ('parity', 'bf16'):
        model.hftt_precision = mode
        ts = TrainStep(model)
        loss = ts.forward_backward(x, *labels)
        torch.cuda.synchronize()
        eng = ts.engine
        res[mode] = ([t.clone() for t in eng._ws[B]['outs']], loss[0].item(),
                     {name: eng.flat_grads[o:o + n].clone() for (name, _, o, n) in eng._bound})
        del ts
        eng._ws.clear()
        torch.cuda.empty_cache()
    rep = {}
    for n, a, b in zip(OUT_NAMES, res['bf16'][0], res['parity'][0]):
        rep[n] = max_err(a, b)
    for i in (2, 7):
        a, b = res['bf16'][0][i] >= 0.5, res['parity'][0][i] >= 0.5
        tp = (a & b).sum().item(); fp = (a & ~b).sum().item(); fn = (~a & b).sum().item()
        rep[OUT_NAMES[i] + '.frame_f1'] = 2 * tp / (2 * tp + fp + fn) if (tp + fp + fn) else 1.0
        flip = a != b
        band = (res['parity'][0][i][flip] - 0.5).abs().max().item() if flip.any() else 0.0
        assert band <= rep[OUT_NAMES[i]] + 1e-6
    rep['loss_rel_err'] = abs(res['bf16'][1] - res['parity'][1]) / res['parity'][1]
    cos = {}
    for name, gp in res['parity'][2].items():
        if gp.abs().max().item() < 1e-7 or name.endswith('fc_k.bias'):      # (a key bias shifts every logit of a row alike: its gradient is
            continue                                                         #  identically zero, what is computed is rounding residue)
        first = name.startswith('encoder') and any(t in name for t in ('conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq',
                                                                       'layers_freq.0.self_attention.fc_q', 'layers_freq.0.self_attention.fc_k'))
        cos[name] = (_cos(res['bf16'][2][name], gp), first)
    _report_cosines(rep, cos)
    print('bf16 mode vs parity mode, paper B=8:', json.dumps(rep))
    for n in OUT_NAMES:
        assert rep[n] <= BUDGET[n], (n, rep[n])
    assert rep['mpe_A.frame_f1'] > 0.98 and rep['mpe_B.frame_f1'] > 0.98
    assert rep['loss_rel_err'] < LOSS_REL
    _assert_cosines(rep, full=True)


def test_x3_mode_paper_b8_against_fp32_mfma_mode(dev):
    """BASELINE config 3 (paper size, batch 8) in the BENCHMARKED mode 'x3' (split operands, three bf16-rate MFMA passes) against the exact
    fp32-MFMA mode on the same device: every output within north_star's 1e-3, no thresholded frame decision outside that band, the same loss,
    and every gradient tensor -- including the ill-conditioned ones behind the first encoder layer's attention (logits ~1e5, where the
    single-pass bf16 mode is ANTI-correlated: -0.85 ... -0.09) -- pointing the same way: cosine >= 0.99."""
    from hftt_hip.trainer import TrainStep
    cfg, B = O.PAPER, 8
    model = util.build_model(cfg, 2024)
    util.perturb(model, 2025)
    model = model.to(dev)
    model.train()
    x = O.synth_spec(B, cfg, salt=31).to(dev)
    labels = _to_dev(O.synth_labels(B, cfg, salt=32), dev)
    res = {}
    for mode in ('parity', 'x3'):
        model.hftt_precision = mode
        ts = TrainStep(model)
        loss = ts.forward_backward(x, *labels)
        torch.cuda.synchronize()
        eng = ts.engine
        res[mode] = ([t.clone() for t in eng._ws[B]['outs']], loss[0].item(),
                     {name: eng.flat_grads[o:o + n].clone() for (name, _, o, n) in eng._bound})
        del ts
        eng._ws.clear()
        torch.cuda.empty_cache()
    rep = {}
    for n, a, b in zip(OUT_NAMES, res['x3'][0], res['parity'][0]):
        rep[n] = max_err(a, b)
        assert rep[n] < 1e-3, (n, rep[n])
    for i in (2, 7):
        a, b = res['x3'][0][i] >= 0.5, res['parity'][0][i] >= 0.5
        tp = (a & b).sum().item(); fp = (a & ~b).sum().item(); fn = (~a & b).sum().item()
        rep[OUT_NAMES[i] + '.frame_f1'] = 2 * tp / (2 * tp + fp + fn) if (tp + fp + fn) else 1.0
        assert rep[OUT_NAMES[i] + '.frame_f1'] > 0.9999
    rep['loss_rel_err'] = abs(res['x3'][1] - res['parity'][1]) / res['parity'][1]
    assert rep['loss_rel_err'] < 1e-5
    cos = {}
    for name, gp in res['parity'][2].items():
        if gp.abs().max().item() < 1e-7 or name.endswith('fc_k.bias'):      # (a key bias shifts every logit of a row alike: its gradient is
            continue                                                         #  identically zero, what is computed is rounding residue)
        first = name.startswith('encoder') and any(t in name for t in ('conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq',
                                                                       'layers_freq.0.self_attention.fc_q', 'layers_freq.0.self_attention.fc_k'))
        cos[name] = (_cos(res['x3'][2][name], gp), first)
    _report_cosines(rep, cos)
    print('x3 mode vs fp32-MFMA mode, paper B=8:', json.dumps(rep))
    worst = min(c for c, _ in cos.values())
    assert worst >= 0.99, sorted((round(c, 4), n) for n, (c, f) in cos.items())[:5]
    assert all(c >= 0.99 for c, f in cos.values() if f)


def test_config5_paper_size_inference(dev, tmp_path):
    """BASELINE config 5 on one GPU: 60 s synthetic plucked-string audio (corpus/synth_audio.py, seed 1234) -> HIP log-mel -> 30 paper-size
    clips through AMT.transcript in the DEFAULT, benchmarked mode (x3): deterministic, every posterior within north_star's 1e-3 of the exact-fp32
    mode on the same features and the same frame decisions; the bf16 throughput mode inside its own (wider) band; decodes to a MIDI file."""
    from model.amt import AMT
    from corpus import synth_audio as SA
    cfg = O.PAPER
    model = util.build_model(cfg, 1234)
    util.perturb(model, 1235)
    f = tmp_path / 'model.pkl'
    with open(f, 'wb') as fh:
        pickle.dump(model, fh, protocol=4)
    amt = AMT(SA.default_config(), str(f), batch_size=32)
    notes = SA.pluck_notes(1234)
    feat = amt.wave2feature(SA.pluck_wave(notes).unsqueeze(0), SA.SR)
    assert feat.shape == (3751, 256)
    names = ['onset_A', 'offset_A', 'mpe_A', 'velocity_A', 'onset_B', 'offset_B', 'mpe_B', 'velocity_B']
    amt.model.hftt_precision = 'parity'
    ref = amt.transcript(feat.numpy())
    res = {}
    for mode in ('x3', 'bf16'):
        amt.model.hftt_precision = mode
        outs = amt.transcript(feat.numpy())
        again = amt.transcript(feat.numpy())
        for a, b in zip(outs, again):
            assert a.shape == (3840, 88) and np.array_equal(a, b)          # 30 clips, bit-deterministic
        rep = {}
        for k, (a, b) in enumerate(zip(outs, ref)):
            if k % 4 == 3:
                rep[names[k] + '.argmax_differs'] = float((a != b).mean())
            else:
                rep[names[k]] = float(np.abs(a - b).max())
                rep[names[k] + '.mean'] = float(np.abs(a - b).mean())
                rep[names[k] + '.decisions_differ'] = int(((a >= 0.5) != (b >= 0.5)).sum())
        res[mode] = (outs, rep)
        print('config 5, paper size, %s vs parity mode:' % mode, json.dumps(rep))
    outs, rep = res['x3']
    for k, nm in enumerate(names):
        if k % 4 == 3:
            assert rep[nm + '.argmax_differs'] < 2e-3, nm          # velocity argmax: near-ties between adjacent classes only
        else:
            assert rep[nm] <= 1e-3 and rep[nm + '.decisions_differ'] == 0, (nm, rep[nm])        # north_star: 1e-3 on the posteriors
    _, rep = res['bf16']
    for k, nm in enumerate(names):
        if k % 4 == 3:
            assert rep[nm + '.argmax_differs'] < 0.08, nm
        else:
            # harmonic audio over a silent floor is harder than noise-like clips: most bins sit at the -18.42 floor, the first encoder
            # layer's attention logits reach ~1e4 and a bf16 product (2^-9 relative) re-ranks near-tied keys.  Worst element / mean:
            assert rep[nm] <= 0.15 and rep[nm + ".mean"] <= 5e-3, (nm, rep[nm], rep[nm + '.mean'])
    est = amt.mpe2note(a_onset=outs[4], a_offset=outs[5], a_mpe=outs[6], a_velocity=outs[7])
    mid = tmp_path / 'out.mid'
    amt.note2midi(est, str(mid))
    assert mid.read_bytes()[:4] == b'MThd'


def test_config5_with_trained_weights(dev, tmp_path):
    """BASELINE config 5 with weights that mean something: tests/golden/config5_tiny_trained.pkl is the reference's default-size model
    (d = 64, 2 + 2 layers: training/m_training.py:56-61) after six minutes of THIS path's training step on the synthetic plucked-string corpus
    (tools/train_config5.py; log: profiles/r04_config5_tiny_trained.json).  The scored minute (seed 1234) is not in the training set.
    Asserted: note-F1 (onset within 50 ms, evaluation/m_transcription.py's criterion) and frame-F1 (mpe >= 0.5, evaluation/m_mpe.py:101)
    against the GENERATING notes, and -- SURVEY section 7's caveat 'trained weights may behave differently' -- that on trained weights the default
    x3 mode keeps every posterior within 1e-3 of the exact-fp32 mode with identical frame decisions, the bf16 mode its frame-F1."""
    from model.amt import AMT
    from corpus import synth_audio as SA
    from evaluation.metrics import note_metrics, frame_metrics
    pkl = os.path.join(util.ROOT, 'tests', 'golden', 'config5_tiny_trained.pkl')
    notes = SA.pluck_notes(1234)
    wave = SA.pluck_wave(notes)
    mpe, rep = {}, {}
    for mode in ('parity', 'x3', 'bf16'):
        amt = AMT(SA.default_config(), pkl, batch_size=32)
        amt.model.hftt_precision = mode
        feat = amt.wave2feature(wave.unsqueeze(0), SA.SR)
        outs = amt.transcript(feat.numpy())
        est = amt.mpe2note(a_onset=outs[4], a_offset=outs[5], a_mpe=outs[6], a_velocity=outs[7])
        nm = note_metrics(notes, est)
        fm = frame_metrics(SA.reference_roll(notes, feat.shape[0]), outs[6], threshold=0.5)
        mpe[mode] = outs
        rep[mode] = {'note_f1': round(nm['F-measure'], 4), 'frame_f1': round(fm['f1'], 4), 'n_est': len(est)}
    print('config 5 with trained weights:', json.dumps(rep))
    assert rep['x3']['note_f1'] > 0.90 and rep['x3']['frame_f1'] > 0.85, rep
    assert rep['bf16']['note_f1'] > 0.90 and abs(rep['bf16']['frame_f1'] - rep['parity']['frame_f1']) < 5e-3, rep
    for k in (0, 1, 2, 4, 5, 6):
        a, b = mpe['x3'][k], mpe['parity'][k]
        assert float(np.abs(a - b).max()) <= 1e-3 and int(((a >= 0.5) != (b >= 0.5)).sum()) == 0, k
    assert float((mpe['x3'][7] != mpe['parity'][7]).mean()) < 2e-3 and float((mpe['x3'][3] != mpe['parity'][3]).mean()) < 2e-3


def _paper_trained_pkl(tmp_path):
    """tests/golden/config5_paper_trained.npz (tools/pack_checkpoint.py) -> this repo's module -> a pickle as m_training.py:372-373 writes it"""
    sys.path.insert(0, os.path.join(util.ROOT, 'tools'))
    from pack_checkpoint import unpack_state_dict
    sd = unpack_state_dict(np.load(os.path.join(util.GOLDEN, 'config5_paper_trained.npz')))
    model = util.build_model(O.PAPER, 1)
    model.load_state_dict(sd)
    f = tmp_path / 'paper_trained.pkl'
    with open(f, 'wb') as fh:
        pickle.dump(model, fh, protocol=4)
    return model, str(f)


def test_config5_paper_size_trained_weights_against_the_reference(dev, tmp_path):
    """BASELINE config 5 at PAPER size with weights that mean something.  tests/golden/config5_paper_trained.npz is the
    paper-size model (d 256, ff 512, 3+3 layers, 4 heads) after 12,000 steps of THIS path's training step in the default x3 mode on the synthetic
    plucked-string corpus (six minutes at 266 clips/s; recipe and log: profiles/r05_config5_paper_trained.json).
    (1) Against the REFERENCE module's own CPU outputs on those weights (config5_paper_golden.npz, tests/golden/make_golden_r5.py: two of the
        minute's 30 clips): every posterior and logit within north_star's 1e-3, no frame decision different, the velocity argmax equal wherever
        the reference's top-two margin exceeds 2e-3, the attention map within 1e-3 -- the default mode against the reference, not against
        another mode of this repo.
    (2) The whole of config 5 from the HIP log-mel on: note-F1 (onset, 50 ms) and frame-F1 against the GENERATING notes of the unseen minute
        (seed 1234), and x3 against the exact-fp32 mode on all 30 clips."""
    from model.amt import AMT
    from corpus import synth_audio as SA
    from evaluation.metrics import note_metrics, frame_metrics
    model, pkl = _paper_trained_pkl(tmp_path)
    g = util.golden('config5_paper_golden')
    # ---- (1) the reference's outputs ----
    m = model.to(dev).eval()
    m.hftt_precision = 'x3'
    with torch.no_grad():
        out = [t.float().cpu() for t in m(torch.from_numpy(g['input']).to(dev))]
    rep = {}
    for n, t in zip(util.OUT_NAMES, out):
        if 'out.' + n in g.files:
            ref = torch.from_numpy(g['out.' + n])
            rep[n] = max_err(t, ref)
            assert rep[n] <= 1e-3, (n, rep[n])
            assert int(((t >= 0.5) != (ref >= 0.5)).sum()) == 0, n
        else:
            f = t.reshape(-1)
            rep[n] = max_err(f[::int(g['out.' + n + '.stride'])], torch.from_numpy(g['out.' + n + '.sample']))
            assert rep[n] <= 1e-3, (n, rep[n], float(g['out.' + n + '.stats'][1]))
            if n != 'attention':
                decided = torch.from_numpy(g['out.' + n + '.margin']) > 2e-3
                same = t.argmax(-1) == torch.from_numpy(g['out.' + n + '.argmax']).long()
                assert bool(same[decided].all()), n
                rep[n + '.argmax_equal'] = float(same.float().mean())
    print('config 5, paper size, trained weights, x3 vs the reference (2 clips):', json.dumps({k: float('%.3g' % v) for k, v in rep.items()}))
    # ---- (2) the whole minute ----
    notes = SA.pluck_notes(1234)
    wave = SA.pluck_wave(notes)
    res, outs = {}, {}
    for mode in ('parity', 'x3', 'bf16'):
        amt = AMT(SA.default_config(), pkl, batch_size=32)
        amt.model.hftt_precision = mode
        feat = amt.wave2feature(wave.unsqueeze(0), SA.SR)
        o = amt.transcript(feat.numpy())
        est = amt.mpe2note(a_onset=o[4], a_offset=o[5], a_mpe=o[6], a_velocity=o[7])
        nm = note_metrics(notes, est)
        fm = frame_metrics(SA.reference_roll(notes, feat.shape[0]), o[6], threshold=0.5)
        outs[mode] = o
        res[mode] = {'note_f1': round(nm['F-measure'], 4), 'note_P': round(nm['Precision'], 4), 'note_R': round(nm['Recall'], 4),
                     'frame_f1': round(fm['f1'], 4), 'n_est': len(est), 'n_ref': len(notes)}
    print('config 5, paper size, trained weights, against the generating notes:', json.dumps(res))
    assert res['x3']['note_f1'] >= 0.90 and res['x3']['frame_f1'] >= 0.85, res
    assert res['bf16']['note_f1'] >= 0.90 and abs(res['bf16']['frame_f1'] - res['parity']['frame_f1']) < 5e-3, res
    for k in (0, 1, 2, 4, 5, 6):
        a, b = outs['x3'][k], outs['parity'][k]
        assert float(np.abs(a - b).max()) <= 1e-3 and int(((a >= 0.5) != (b >= 0.5)).sum()) == 0, k
