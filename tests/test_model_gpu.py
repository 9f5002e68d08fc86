"""End-to-end parity of the HIP path (through model.model_spec2midi / the C ABI) against the CPU oracle and the golden
fixtures generated from the reference.  Tolerance of the parity mode: 1e-3 max-abs on the 8 posterior/logit outputs
(BASELINE.json north_star); observed errors are ~1e-4 and are printed for the record."""
import io
import json
import os
import pickle

import numpy as np
import pytest
import torch

import util
from util import keep_scale, O, MINI, OUT_NAMES, max_err

pytestmark = pytest.mark.gpu
TOL_OUT = 1e-3


def _to_dev(labels, dev):
    return tuple(t.to(dev).contiguous() for t in labels)


def _oracle_run(model_cpu_sd, cfg, x, labels, wA=1.0, wB=1.0):
    sd = {k: v.clone().requires_grad_(True) for k, v in model_cpu_sd.items()}
    out = O.model_forward(sd, x, cfg)
    loss = O.spec2midi_loss(out, *labels, wA, wB)
    loss.backward()
    return out, loss.item(), {k: v.grad for k, v in sd.items()}


def _grad_check(eng, names, ref_grads, tol_rel, report):
    worst = 0.0
    for (name, _, o, n) in eng._bound:
        g = eng.flat_grads[o:o + n].view(eng.pshape[name]).cpu()
        assert torch.isfinite(g).all(), 'gradient of %s not written / not finite' % name
        ref = ref_grads[name]
        scale = max(ref.abs().max().item(), 1e-6)
        err = (g.double() - ref.double()).abs().max().item()
        if ref.abs().max().item() < 1e-7:      # fc_k.bias: exactly-zero gradient up to rounding noise
            assert err < 1e-5, name
            continue
        worst = max(worst, err / scale)
        assert err / scale < tol_rel, (name, err, scale)
    report['worst_grad_rel'] = worst


@pytest.mark.parametrize('precision', ['x3', 'parity'])
def test_mini_full_tensors_and_all_grads(dev, precision):
    """Every output element, the loss and every parameter gradient at a small config (oracle pinned by golden/micro)."""
    from hftt_hip.trainer import TrainStep
    cfg, B = MINI, 3
    model = util.build_model(cfg, 77)
    util.perturb(model, 78)
    sd = util.sd_cpu(model)
    x = O.synth_spec(B, cfg, salt=5) * 0.5
    labels = O.synth_labels(B, cfg, salt=6)
    ref_out, ref_loss, ref_grads = _oracle_run(sd, cfg, x, labels, 1.0, 0.7)
    model = model.to(dev)
    model.hftt_precision = precision
    model.train()                               # dropout 0.0: identical to eval
    ts = TrainStep(model, weight_A=1.0, weight_B=0.7)
    ts.engine.flat_grads.fill_(float('nan'))
    loss = ts.forward_backward(x.to(dev), *_to_dev(labels, dev))
    outs = ts.engine._ws[B]['outs']
    rep = {}
    for n, t, r in zip(OUT_NAMES, outs, ref_out):
        assert t.shape == r.shape
        rep[n] = max_err(t, r)
        assert rep[n] < TOL_OUT, (n, rep[n])
    assert abs(loss[0].item() - ref_loss) < 1e-4
    _grad_check(ts.engine, None, ref_grads, 2e-3, rep)
    print('mini', precision, json.dumps(rep))
    # eval mode gives the same outputs; the autograd (compat) path gives the same gradients as the fast path
    model.eval()
    with torch.no_grad():
        ev = model(x.to(dev))
    for a, b in zip(ev, outs):
        assert max_err(a, b) == 0.0
    model.train()
    fast = ts.engine.flat_grads.clone()
    out2 = model(x.to(dev))
    assert out2[0].requires_grad and not out2[4].requires_grad
    l2 = O.spec2midi_loss(out2, *_to_dev(labels, dev), 1.0, 0.7)       # torch loss on device outputs (compat path)
    l2.backward()
    flat2 = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    flat1 = torch.cat([fast[o:o + n] for (_, _, o, n) in ts.engine._bound])
    # (x3: the gradient products run on bf16 pairs, 2^-16 per product -- the two paths differ in the last bits of the incoming output
    # gradients, fused loss kernel vs torch autograd, and that much comes back out)
    assert (flat1 - flat2).abs().max().item() < (3e-4 if precision == 'x3' else 1e-5) * max(1.0, flat1.abs().max().item())
    # one fused Adam step == oracle Adam step
    names = [n for n, _ in model.named_parameters()]
    ts.engine.flat_grads.copy_(fast)
    ts.opt.step()
    params = [sd[k].clone() for k in names]
    grads = [ref_grads[k] for k in names]
    O.adam_step(params, grads, [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params], 1, lr=1e-4)
    for k, p, gr, mine in zip(names, params, grads, model.parameters()):
        # step 1 of Adam moves every element by lr * g/(|g| + eps): elements with |g| ~ eps amplify rounding noise
        solid = gr.abs() > (1e-6 if precision == 'parity' else max(1e-6, 2e-2 * gr.abs().max().item()))      # x3: gradients carry ~2e-4 of the tensor's largest
        diff = (mine.detach().cpu().double() - p.double()).abs()
        assert diff[solid].max().item() < 3e-6 if solid.any() else True, k
        assert diff.max().item() <= 2.0e-4 + 1e-7, k


@pytest.mark.parametrize('precision', ['x3', 'parity'])
@pytest.mark.parametrize('name', ['tiny_b2', 'paper_b1'])
def test_golden_fixture(dev, name, precision):
    """Reference outputs / loss / gradient statistics recorded in tests/golden (no oracle in the loop), at north_star's 1e-3, in both modes
    that claim it: 'x3' (split fp16 / bf16 operands, three bf16-rate MFMA passes) and 'parity' (exact fp32 MFMA)."""
    from hftt_hip.trainer import TrainStep
    g = util.golden(name)
    cfg = util.cfg_from_golden(g)
    seed, B = int(g['seed']), int(g['bsz'])
    model = util.build_model(cfg, seed)
    util.perturb(model, seed + 1)
    model = model.to(dev)
    model.hftt_precision = precision
    model.train()
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
        rep[n] = max_err(t.reshape(-1)[::st], ref)
        assert rep[n] < TOL_OUT, (n, rep[n])
        assert abs(t.double().sum().item() - g['out.' + n + '.stats'][0]) < 1e-3 * max(1.0, abs(g['out.' + n + '.stats'][0])) + 1e-2 * t.numel() ** 0.5
    assert abs(loss[0].item() - float(g['loss'])) < 2e-4
    worst = 0.0
    for (pname, _, o, n) in ts.engine._bound:
        gr = ts.engine.flat_grads[o:o + n].cpu()
        assert torch.isfinite(gr).all(), pname
        stats = g['grad.' + pname + '.stats']
        st = max(1, n // 64) | 1
        ref = torch.from_numpy(g['grad.' + pname + '.sample'])
        if stats[1] < 1e-7:
            continue
        e = max_err(gr[::st], ref) / stats[1]
        worst = max(worst, e)
        # Gradients that flow through the FIRST encoder layer's attention are ill-conditioned on raw log-mel input (logits
        # ~1e5, near one-hot softmax): two correct fp32 implementations differ there by ~1e-3..1e-2 relative -- against an fp64
        # evaluation the device is at 2.5e-3 and a CPU fp32 run at 4e-3 on this input (test_parity_gradients_against_fp64_evaluation),
        # the recorded reference run (another host, another BLAS blocking) at ~1.2e-2.  Everything downstream of the first LayerNorm is tight.
        first = any(t in pname for t in ('encoder_spec2midi.conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq',
                                         'layers_freq.0.self_attention.fc_q', 'layers_freq.0.self_attention.fc_k')) and pname.startswith('encoder')
        if first:
            rep['first_layer_worst'] = max(rep.get('first_layer_worst', 0.0), e)
        assert e < (2e-2 if first else 5e-3), (pname, e)
        assert abs(gr.double().norm().item() - stats[2]) < (2e-2 if first else 2e-3) * stats[2], pname
    rep['worst_grad_rel'] = worst
    print(name, precision, json.dumps(rep))


_OTHER = dict(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, n_note=8, n_velocity=16)
OTHER_CONFIGS = {'d256_ff1024': O.HfttConfig(hid_dim=256, pf_dim=1024, enc_layer=1, dec_layer=2, enc_head=4, dec_head=4, **_OTHER),      # x3: block GEMMs at the strip width
                 'd128_ff256': O.HfttConfig(hid_dim=128, pf_dim=256, enc_layer=2, dec_layer=1, enc_head=4, dec_head=2, **_OTHER),
                 # the reference's default width (m_training.py:56-61): x3 runs it on the small-width strip family (csrc/x3s_strip.h)
                 'd64_ff128': O.HfttConfig(hid_dim=64, pf_dim=128, enc_layer=2, dec_layer=2, enc_head=2, dec_head=2, **_OTHER),
                 'd256_ff512': O.HfttConfig(hid_dim=256, pf_dim=512, enc_layer=1, dec_layer=1, enc_head=4, dec_head=4, **_OTHER),        # strip plans, odd batch
                 # 15 frames x 8 notes x batch 3 = 360 note tokens: not a multiple of the strip kernels' 32 -> this workspace falls back to the block plans
                 'd256_ff512_t15': O.HfttConfig(hid_dim=256, pf_dim=512, enc_layer=1, dec_layer=2, enc_head=4, dec_head=4, **dict(_OTHER, n_frame=15))}


@pytest.mark.parametrize('precision', ['x3', 'parity'])
@pytest.mark.parametrize('name', sorted(OTHER_CONFIGS))
def test_other_configurations_forward_and_gradients(dev, name, precision):
    """Shapes the fixtures do not cover (the split-operand block GEMMs at d = 256 with ff = 1024, d = 128 with dh = 32 / 64 heads, the strip
    plans at a batch of 3): every output within north_star's 1e-3 of the CPU oracle.  Gradients: within 5e-3 of each tensor's maximum -- at d = 256
    the tensors behind the first encoder layer's attention (logits ~1e4 on raw log-mel x sqrt(d): a handful of near-tied softmax rows decide
    them) are 1.1e-3 .. 1.4e-3 off in the EXACT fp32 mode already (fp32 summation order), x3 1.4e-3 .. 2.9e-3; every other
    tensor is below 1.1e-3 in both modes (d = 128: 5e-4 everywhere)."""
    from hftt_hip.trainer import TrainStep
    cfg, B = OTHER_CONFIGS[name], 3
    x = O.synth_spec(B, cfg, salt=31)
    labels = O.synth_labels(B, cfg, salt=32)
    model = util.build_model(cfg, 2024, dropout=0.0)
    ref_out, ref_loss, ref_grads = _oracle_run({k: v.detach().clone() for k, v in model.state_dict().items()}, cfg, x, labels)
    model = model.to(dev)
    model.hftt_precision = precision
    model.train()
    ts = TrainStep(model)
    loss = ts.forward_backward(x.to(dev), *_to_dev(labels, dev))
    torch.cuda.synchronize()
    eng = ts.engine
    rep = {n: max_err(o, r.detach()) for n, o, r in zip(OUT_NAMES, eng._ws[B]['outs'], ref_out)}
    print(name, precision, 'strip' if eng._ws[B]['strip'] else 'block', json.dumps({k: float('%.2g' % v) for k, v in rep.items()}))
    for n, e in rep.items():
        assert e < TOL_OUT, (n, e)
    if name == 'd64_ff128' and precision == 'x3':
        assert eng.strip_small and eng._ws[B]['strip'] and any('x3s_mlp_kernel' in (m or {}).get('kernel', '') for _, _, _, m in eng._ws[B]['fwd'])
    if name == 'd256_ff512_t15':
        assert eng._ws[B]['strip'] is False and (eng.strip or precision != 'x3')      # x3: strip engine, block plans for this token count
    assert abs(loss[0].item() - ref_loss) < 1e-4 * abs(ref_loss)
    errs = []
    for (pname, _, o, n) in eng._bound:
        g = eng.flat_grads[o:o + n].view(eng.pshape[pname]).cpu().double()
        ref = ref_grads[pname].double()
        if ref.abs().max().item() < 1e-7:
            continue
        errs.append(((g - ref).abs().max().item() / ref.abs().max().item(), pname))
    errs.sort(reverse=True)
    first = ('conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq', 'layers_freq.0.self_attention.fc_q', 'layers_freq.0.self_attention.fc_k')
    rest = [e for e, n_ in errs if not (n_.startswith('encoder') and any(t in n_ for t in first))]
    print('  worst gradient errors / max:', [(float('%.2g' % e), n_) for e, n_ in errs[:3]], 'beyond the first layer:', float('%.2g' % max(rest)))
    assert errs[0][0] < 5e-3 and max(rest) < 2e-3


def test_parity_gradients_against_fp64_evaluation(dev):
    """Paper size, B = 1: every parity-mode gradient against an fp64 evaluation of the oracle graph, with the same graph in CPU fp32 as the
    yardstick.  The tensors behind the first encoder layer's attention (logits ~1e5, 99.6 % of the rows one-hot to 1e-6) are where any fp32
    implementation is noisy: a CPU fp32 run is 3-4e-3 from fp64 there.  Since the fp32 MFMA GEMM sums each 32-wide k stage from zero before adding
    it to the running sum (gemm_nt.hip), the device sits at the CPU's level (one running accumulator: 6x above it, 2.5e-2)."""
    from hftt_hip.trainer import TrainStep
    cfg, B, seed = O.PAPER, 1, 2024
    model = util.build_model(cfg, seed)
    util.perturb(model, seed + 1)
    x = O.synth_spec(B, cfg, salt=seed)
    labels = O.synth_labels(B, cfg, salt=seed + 7)
    sd = util.sd_cpu(model)

    def oracle_grads(dt):
        p = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
        O.spec2midi_loss(O.model_forward(p, x.to(dt), cfg), *labels).backward()
        return {k: v.grad.double().reshape(-1) for k, v in p.items() if v.grad is not None}

    g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
    model = model.to(dev)
    model.hftt_precision = 'parity'
    model.train()
    ts = TrainStep(model)
    ts.forward_backward(x.to(dev), *_to_dev(labels, dev))
    worst, worst_first, worst_ratio, rows = 0.0, 0.0, 0.0, []
    for (pname, _, o, n) in ts.engine._bound:
        ref = g64[pname]
        sc = ref.abs().max().item()
        if sc < 1e-7:
            continue                                  # (fc_k biases: exactly zero gradient in exact arithmetic)
        e_dev = (ts.engine.flat_grads[o:o + n].cpu().double() - ref).abs().max().item() / sc
        e_cpu = (g32[pname] - ref).abs().max().item() / sc
        first = any(t in pname for t in ('encoder_spec2midi.conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq',
                                         'layers_freq.0.self_attention.fc_q', 'layers_freq.0.self_attention.fc_k')) and pname.startswith('encoder')
        if first:
            worst_first = max(worst_first, e_dev)
            worst_ratio = max(worst_ratio, e_dev / max(e_cpu, 1e-12))
            rows.append((pname, e_dev, e_cpu))
        else:
            worst = max(worst, e_dev)
            assert e_dev < 5e-3, (pname, e_dev, e_cpu)
    for r in rows:
        print('  %-66s device %.2e  cpu fp32 %.2e' % r)
    print('parity gradients vs fp64 (paper B=1): others worst %.2e, first-layer worst %.2e (at most %.2f x the CPU fp32 error)' % (worst, worst_first, worst_ratio))
    # the noise of these tensors is heavy-tailed (a handful of near-tied rows decide it) and moves with the seed: 2.5e-3 .. 9e-3 on the device,
    # 3e-3 .. 5e-3 on the CPU.  Bound: a small multiple of the CPU's own error (it was 6x before the two-level sum), and 1.5e-2 absolute.
    assert worst_ratio < 3.0 and worst_first < 1.5e-2, (worst_ratio, worst_first)


def test_bf16_mode_error_and_frame_f1(dev):
    """Throughput mode (single-pass bf16 MFMA): report its error; thresholded frame decisions must agree with parity mode."""
    cfg, B = O.TINY, 2
    model = util.build_model(cfg, 4321)
    util.perturb(model, 4322)
    model = model.to(dev).eval()
    x = O.synth_spec(B, cfg, salt=4321).to(dev)
    with torch.no_grad():
        model.hftt_precision = 'parity'
        ref = [t.clone() for t in model(x)]
        model.hftt_precision = 'bf16'
        out = model(x)
    rep = {n: max_err(a, b) for n, a, b in zip(OUT_NAMES, out, ref)}
    print('bf16-mode error vs parity mode (tiny):', json.dumps(rep))
    for n in ('onset_A', 'offset_A', 'mpe_A', 'onset_B', 'offset_B', 'mpe_B', 'attention'):
        assert rep[n] < 0.1
    for i in (2, 7):          # frame decisions mpe >= 0.5 (evaluation/m_mpe.py:101)
        a, b = (out[i] >= 0.5), (ref[i] >= 0.5)
        tp = (a & b).sum().item(); fp = (a & ~b).sum().item(); fn = (~a & b).sum().item()
        f1 = 2 * tp / max(1, 2 * tp + fp + fn) if (tp + fp + fn) else 1.0
        flipped = (a != b).float().mean().item()
        # untrained weights leave many posteriors within the bf16 error band of the 0.5 threshold: every flipped decision
        # must be explained by that band, and flips must be rare
        band = (ref[i][a != b] - 0.5).abs().max().item() if (a != b).any() else 0.0
        print('bf16 mode %s: frame-F1 vs parity mode %.4f, flipped decisions %.4f%%, widest flipped margin %.4f' % (OUT_NAMES[i], f1, 100 * flipped, band))
        assert band <= rep[OUT_NAMES[i]] + 1e-6
        assert flipped < 0.05


def test_dropout_training_mode(dev):
    """Dropout on (the reference's training default 0.1): masks differ per step, eval is deterministic, loss is in family
    with the dropout-on CPU restatement, and a few Adam steps reduce the loss."""
    from hftt_hip.trainer import TrainStep
    cfg, B = MINI, 4
    model = util.build_model(cfg, 11, dropout=0.1)
    sd = util.sd_cpu(model)
    x = O.synth_spec(B, cfg, salt=9) * 0.5
    labels = O.synth_labels(B, cfg, salt=10)
    torch.manual_seed(0)
    ref_losses = []
    for _ in range(8):
        with torch.no_grad():
            ref_losses.append(O.spec2midi_loss(O.model_forward(sd, x, cfg, p=0.1, training=True), *labels).item())
    model = model.to(dev)
    model.train()
    ts = TrainStep(model, lr=1e-3)
    xd, ld = x.to(dev), _to_dev(labels, dev)
    losses = [ts.forward_backward(xd, *ld)[0].item() for _ in range(8)]
    assert len(set(round(l, 6) for l in losses)) > 1                       # different masks every step
    assert abs(np.mean(losses) - np.mean(ref_losses)) < 4 * (np.std(ref_losses) + np.std(losses)) / np.sqrt(8) + 0.05
    assert torch.isfinite(ts.engine.flat_grads).all()
    first = np.mean(losses)
    for _ in range(30):
        last = ts(xd, *ld)[0].item()
    assert last < first - 0.1
    model.eval()
    with torch.no_grad():
        a = model(xd); b = model(xd)
    assert all(max_err(p, q) == 0.0 for p, q in zip(a, b))


# d = 256 / ff = 512 (the paper's width): the strip plans.  x3_256: 32 keys per attention (plane forms at two workgroups per CU);
# x3_256_long: 256 bins -- the 256-key plane kernels bench.py's step is made of (x3p_attn_fwd_kernel<8, 8, ...>, the persistent staggered
# x3_attn_bwd_kernel<8, 64, true, 1> = the roofline kernel, and the cross-attention forms with 40 notes = two query blocks per item)
_WIDE = dict(n_margin=4, cnn_channel=4, cnn_kernel=5, hid_dim=256, pf_dim=512, enc_layer=1, dec_layer=2, enc_head=4, dec_head=4, n_velocity=16)
DROPOUT_CASES = {
    'parity': ('parity', MINI), 'parity256': ('parity', O.HfttConfig(n_frame=16, n_bin=32, n_note=8, **_WIDE)),
    'bf16': ('bf16', O.HfttConfig(n_frame=16, n_bin=32, n_note=8, **_WIDE)),
    # the BENCHMARKED mode: its own launch sequence and site wiring -- block plans (MINI: ff = 96), the small-width strip
    # family (the reference's default width, ff = 128), the strip + plane plans at the paper's width
    'x3': ('x3', MINI),
    'x3_small': ('x3', O.HfttConfig(n_margin=4, n_frame=16, n_bin=48, cnn_channel=4, cnn_kernel=5, hid_dim=64, pf_dim=128, enc_layer=2, dec_layer=2,
                                    enc_head=2, dec_head=2, n_note=12, n_velocity=16)),
    'x3_256': ('x3', O.HfttConfig(n_frame=16, n_bin=32, n_note=8, **_WIDE)),
    'x3_256_long': ('x3', O.HfttConfig(n_frame=8, n_bin=256, n_note=40, **_WIDE)),
    # three decoder layers (the paper's depth): the cross-attention K / V projections of all layers as ONE launch (N = 1536) and, in the backward, ONE
    # weight-gradient product with six segments + the encoder-output gradient as two K = 768 products (round 6) -- every gradient, not samples
    'x3_256_dec3': ('x3', O.HfttConfig(n_frame=16, n_bin=32, n_note=8, **dict(_WIDE, dec_layer=3))),
}


def _kernels(plan):
    return [(m or {}).get('kernel', '') for _, _, _, m in plan]


@pytest.mark.parametrize('case', list(DROPOUT_CASES))
def test_dropout_on_outputs_and_gradients_with_the_device_masks_exported_to_the_oracle(dev, monkeypatch, case):
    """The benchmarked training path runs ~40 hash-mask sites, each regenerated in the backward.  Here the oracle is given the SAME masks (its
    dropout calls, in reference order -- model_spec2midi.py:95,236,242,348,372 --, are answered from the numpy emulation of the device generator
    with the engine's seed and site numbers), so every output and every gradient can be compared exactly as in the dropout-free test: a site /
    seed / index mismatch between a forward and a backward launch, or between two kernels of one site, shows up as a gradient error.  The
    control -- the oracle's masks taken from sites shifted by one -- must be far off, or the comparison proves nothing."""
    from hftt_hip.trainer import TrainStep
    precision, cfg = DROPOUT_CASES[case]
    B, p = 2, 0.1
    model = util.build_model(cfg, 31, dropout=p)
    util.perturb(model, 32)
    sd = util.sd_cpu(model)
    x = O.synth_spec(B, cfg, salt=13) * (0.05 if precision == 'bf16' else 0.5)       # (bf16: small logits, so that rounding is not the story)
    labels = O.synth_labels(B, cfg, salt=14)
    model = model.to(dev)
    model.hftt_precision = precision
    model.train()
    ts = TrainStep(model)
    loss = ts.forward_backward(x.to(dev), *_to_dev(labels, dev))
    eng = ts.engine
    strip_on = os.environ.get('HFTT_STRIP', '1') != '0'
    ws = eng._ws[B]
    if precision != 'x3':
        assert eng.strip == (precision == 'bf16' and strip_on)
    elif strip_on:                          # which launch sequence did the x3 case really run?
        fk, bk = _kernels(ws['fwd']), _kernels(ws['bwd'])
        if case == 'x3':
            assert not eng.strip and any(k.startswith('gemm_nt_kernel') for k in fk)
        elif case == 'x3_small':
            assert eng.strip_small and ws['strip'] and any('x3s_mlp_kernel<0' in k for k in fk) and any('x3s_mlp_kernel<1' in k for k in bk)
        else:
            assert eng.strip and ws['strip'] and not eng.strip_small
            # (forward: the FFN block alone, or -- the default since round 6 -- fc_o + LayerNorm + FFN as one launch, hftt_attn_out_ffn_fwd)
            assert any(k.startswith('x3_oln_mlp_kernel<' if eng.fuse_offn_opt == 'all' else 'x3_mlp_kernel<0, 16') for k in fk)
            assert any(k.startswith('x3_mlp_kernel<1, 16') for k in bk)
            assert any(k.startswith('x3p_attn_fwd_kernel') for k in fk)
            if case == 'x3_256_long':       # the kernels of bench.py's step, dropout form 1 (one hash per key quad)
                assert 'x3_attn_bwd_kernel<8, 64, true, 1>' in bk and 'x3p_attn_fwd_kernel<8, 8, false, 1>' in fk
                assert 'x3p_attn_fwd_kernel<8, 4, true, 1>' in fk, fk      # cross attention of the last decoder layer (writes the attention map)
            else:
                assert 'x3_attn_bwd_kernel<1, 64, true, 1>' in bk, bk
            if case == 'x3_256_dec3':
                assert eng.merge_ckv and eng.merge_ckv_bwd
                assert 'x3_linear_n_kernel<2, 48, false, true, false>' in fk, fk           # the stacked K / V projection
                assert sum(k == 'x3_linear_kernel<4, false, 1, 3, true, 8>' for k in bk) >= 2 and 'x3_linear_kernel<4, false, 1, 3, false, 8>' in bk, bk
            elif case in ('x3_256', 'x3_256_long'):
                assert eng.merge_ckv and not eng.merge_ckv_bwd
    seed, n_sites = ws['seed'], eng._site

    def oracle_with_sites(order):
        site = iter(order)

        def drop(t, pp, training):
            if not (training and pp > 0.0):
                return t
            m = util.keep_mask_t(seed, next(site), tuple(t.shape), pp).to(t.dtype)
            return t * m * keep_scale(pp)
        monkeypatch.setattr(O, '_drop', drop)
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out = O.model_forward(sdg, x, cfg, p=p, training=True)
        l = O.spec2midi_loss(out, *labels)
        l.backward()
        assert next(site, None) is None, 'the oracle made fewer dropout calls than the engine has sites'
        return sdg, out, l
    sdg, ref_out, ref_loss = oracle_with_sites(range(1, n_sites + 1))
    outs = ws['outs']
    tol_out, tol_loss, tol_g = (0.3, 5e-2, None) if precision == 'bf16' else (TOL_OUT, 1e-3, 3e-2)      # bf16 vs an fp32 oracle at random init: DESIGN.md section 2
    worst = max(max_err(outs[k], ref_out[k]) for k in (0, 1, 2, 5, 6, 7))
    assert worst < tol_out, worst
    assert abs(loss[0].item() - ref_loss.item()) < tol_loss * max(1.0, abs(ref_loss.item()))
    if precision == 'x3':
        assert max(max_err(outs[k], ref_out[k]) for k in (3, 8)) < TOL_OUT and max_err(outs[4], ref_out[4]) < TOL_OUT      # velocity logits, attention map
    if precision != 'bf16':
        rep = {}
        _grad_check(eng, None, {k: v.grad for k, v in sdg.items()}, tol_g, rep)
        # control: masks from the WRONG sites (shifted by one) -- the same check must fail, by a wide margin
        sdw, _, _ = oracle_with_sites(list(range(2, n_sites + 1)) + [1])
        wrong = 0.0
        for (name, _, o, n) in eng._bound:
            ref = sdw[name].grad
            if ref.abs().max().item() >= 1e-7:
                wrong = max(wrong, (eng.flat_grads[o:o + n].view(eng.pshape[name]).cpu().double() - ref.double()).abs().max().item() / ref.abs().max().item())
        print('dropout-on %s gradients (%d sites): worst relative error %.2e, posteriors %.2e; with shifted sites %.2e'
              % (case, n_sites, rep['worst_grad_rel'], worst, wrong))
        assert wrong > max(10 * rep['worst_grad_rel'], 10 * tol_g), (wrong, rep)
    else:                                   # bf16: the whole-gradient direction (a wrong mask anywhere turns it)
        g = torch.cat([eng.flat_grads[o:o + n] for (name, _, o, n) in eng._bound if not name.endswith('fc_k.bias')]).double().cpu()
        r = torch.cat([sdg[name].grad.reshape(-1) for (name, _, o, n) in eng._bound if not name.endswith('fc_k.bias')]).double()
        cos = float((g @ r) / (g.norm() * r.norm()))
        # control: the same comparison with the oracle's masks taken from the WRONG sites (shifted by one) must be far off
        site2 = iter(list(range(2, n_sites + 1)) + [1])
        monkeypatch.setattr(O, '_drop', lambda t, pp, training: t if not (training and pp > 0.0) else
                            t * util.keep_mask_t(seed, next(site2), tuple(t.shape), pp).to(t.dtype) * keep_scale(pp))
        sdw = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        O.spec2midi_loss(O.model_forward(sdw, x, cfg, p=p, training=True), *labels).backward()
        w = torch.cat([sdw[name].grad.reshape(-1) for (name, _, o, n) in eng._bound if not name.endswith('fc_k.bias')]).double()
        cos_wrong = float((g @ w) / (g.norm() * w.norm()))
        print('dropout-on bf16 (strip) gradients: cosine with the oracle %.4f (with shifted sites: %.4f), posterior max err %.3f' % (cos, cos_wrong, worst))
        # Yardstick: the bf16 build WITHOUT the strip kernels (HFTT_STRIP=0) runs the plan branches the parity case above has just proven
        # consistent, with the same seed and site numbers.  bf16 rounding under dropout costs it a few 1e-2 of cosine at this size (0.95; 0.999
        # with dropout off); the strip plans (their own backward branches) must land in the same place, far from the shifted-site control.
        monkeypatch.setenv('HFTT_STRIP', '0')
        model1 = util.build_model(cfg, 31, dropout=p)
        util.perturb(model1, 32)
        model1 = model1.to(dev)
        model1.hftt_precision = 'bf16'
        model1.train()
        ts1 = TrainStep(model1)
        ts1.forward_backward(x.to(dev), *_to_dev(labels, dev))
        monkeypatch.delenv('HFTT_STRIP')
        assert not ts1.engine.strip and ts1.engine._ws[B]['seed'] == seed and ts1.engine._site == n_sites
        g1 = torch.cat([ts1.engine.flat_grads[o:o + n] for (name, _, o, n) in ts1.engine._bound if not name.endswith('fc_k.bias')]).double().cpu()
        cos1 = float((g1 @ r) / (g1.norm() * r.norm()))
        cos_builds = float((g @ g1) / (g.norm() * g1.norm()))
        print('   bf16 build without strip kernels: cosine with the oracle %.4f; strip vs that build %.4f' % (cos1, cos_builds))
        assert cos > cos1 - 0.03 and cos > cos_wrong + 0.05 and cos_builds > 0.93, (cos, cos1, cos_wrong, cos_builds)


@pytest.mark.parametrize('width', [256, 64])
def test_strip_kernels_match_the_round1_kernels_with_dropout_on(dev, monkeypatch, width):
    """(width 64, round 5: the same yardstick for the small-width bf16 family, csrc/bs_strip.hip -- the reference's default model, BASELINE config 2.)
    bf16 mode at the paper's width runs the strip kernels (bf16 activation + gradient streams, fused FFN); HFTT_STRIP=0 runs the
    round-1 kernels (fp32 streams).  Every dropout site must regenerate the same masks in both builds, forward and backward, as the
    exact-fp32 parity mode does (same plan, same sites, same seed).  Yardstick: each bf16 build's deviation from the parity mode.  Two
    bf16 roundings of this model differ visibly by themselves (first-layer attention logits ~1e4), so the strip
    build must simply not be further from the parity mode than the round-1 build is -- with dropout off AND on (a site / index
    mismatch anywhere would be an O(1) change of the masked activations and of every gradient behind them)."""
    from hftt_hip.trainer import TrainStep
    cfg = O.HfttConfig(n_margin=4, n_frame=16, n_bin=48, cnn_channel=4, cnn_kernel=5, hid_dim=width, pf_dim=2 * width,
                       enc_layer=2, dec_layer=2, enc_head=width // 64 if width > 64 else 2, dec_head=width // 64 if width > 64 else 2, n_note=12, n_velocity=16)
    B = 2
    x = (O.synth_spec(B, cfg, salt=21) * 0.5).to(dev)
    ld = _to_dev(O.synth_labels(B, cfg, salt=22), dev)
    skip = ('conv', 'tok_embedding_freq', 'encoder_spec2midi.pos_embedding_freq', 'encoder_spec2midi.layers_freq.0.self_attention')
    rep = {}
    for drop in (0.0, 0.1):
        res = {}
        for build in ('parity', 'round1', 'strip'):
            monkeypatch.setenv('HFTT_STRIP', '1' if build == 'strip' else '0')
            model = util.build_model(cfg, 7, dropout=drop).to(dev)
            model.hftt_precision = 'parity' if build == 'parity' else 'bf16'
            model.train()
            ts = TrainStep(model, lr=1e-3)
            loss = ts.forward_backward(x, *ld)
            torch.cuda.synchronize()
            eng = ts.engine
            assert eng.strip == (build == 'strip') and torch.isfinite(eng.flat_grads).all()
            if build == 'strip' and width == 64:
                assert eng.strip_small and any('bs_mlp_kernel' in (m or {}).get('kernel', '') for _, _, _, m in eng._ws[B]['fwd'])
                assert any('bs_linear_kernel' in (m or {}).get('kernel', '') for _, _, _, m in eng._ws[B]['bwd'])
            grads = torch.cat([eng.flat_grads[o:o + n] for (name, _, o, n) in eng._bound
                               if not name.endswith('fc_k.bias') and not any(t in name for t in skip)]).double()
            res[build] = ([t.clone() for t in eng._ws[B]['outs']], loss[0].item(), grads)
        for build in ('round1', 'strip'):
            o, l, gr = res[build]
            po, pl, pg = res['parity']
            rep['%s p=%.1f' % (build, drop)] = {
                'post': max(max_err(o[k], po[k]) for k in (0, 1, 2, 5, 6, 7)), 'logit': max(max_err(o[k], po[k]) for k in (3, 8)),
                'loss': abs(l - pl), 'gcos': float((gr @ pg) / (gr.norm() * pg.norm()))}
    print('bf16 builds vs parity mode:', json.dumps(rep))
    for drop in ('0.0', '0.1'):
        r1, st = rep['round1 p=' + drop], rep['strip p=' + drop]
        assert st['post'] < 1.6 * r1['post'] + 0.02, (drop, st, r1)
        assert st['logit'] < 1.6 * r1['logit'] + 0.05, (drop, st, r1)
        assert st['loss'] < 1.6 * r1['loss'] + 0.02, (drop, st, r1)
        # whole-gradient cosine against the parity mode: with dropout on it moves by a few 1e-2 from one mask realisation to the next (bf16 rounding
        # of the first layers' large attention logits), so the two bf16 builds are only asked to stay in the same band
        assert st['gcos'] > r1['gcos'] - (0.03 if drop == '0.0' else 0.06), (drop, st, r1)


def test_one_launch_for_attention_output_and_ffn_changes_no_bit_of_a_step(dev, monkeypatch):
    """HFTT_X3_FUSE_OFFN: 'all' (default) joins fc_o + LayerNorm and the FFN behind it into hftt_attn_out_ffn_fwd in both forward plans, 'inference' only
    in the plan that saves nothing, '0' never.  The joined launch is bit-identical to the two (tests/test_x3_gpu.py), so the nine outputs, the loss
    and EVERY gradient of a dropout-on training step, and the outputs of an eval forward, must be bit-identical across the three settings -- at the
    paper's width and depth (three decoder layers: the cross-attention output projections with a broadcast residual in layer zero)."""
    from hftt_hip.trainer import TrainStep
    cfg = O.HfttConfig(n_frame=16, n_bin=32, n_note=8, **dict(_WIDE, enc_layer=2, dec_layer=3))
    B = 2
    x = (O.synth_spec(B, cfg, salt=41) * 0.5).to(dev)
    ld = _to_dev(O.synth_labels(B, cfg, salt=42), dev)
    res = {}
    for mode in ('0', 'inference', 'all'):
        monkeypatch.setenv('HFTT_X3_FUSE_OFFN', mode)
        model = util.build_model(cfg, 9, dropout=0.1).to(dev)
        model.hftt_precision = 'x3'
        model.train()
        ts = TrainStep(model, lr=1e-3)
        loss = ts.forward_backward(x, *ld)
        eng = ts.engine
        n_tr = sum('x3_oln_mlp_kernel' in (m or {}).get('kernel', '') for _, _, _, m in eng._ws[B]['fwd'])
        assert n_tr == (cfg.enc_layer + 2 * cfg.dec_layer if mode == 'all' else 0), (mode, n_tr)
        tr = ([t.clone() for t in eng._ws[B]['outs']], loss[0].item(), eng.flat_grads.clone())
        model.eval()
        with torch.no_grad():
            ev = [t.clone() for t in model(x)]
        n_inf = sum('x3_oln_mlp_kernel' in (m or {}).get('kernel', '') for _, _, _, m in eng._ws[B]['fwd_inf'])
        assert n_inf == (0 if mode == '0' else cfg.enc_layer + 2 * cfg.dec_layer), (mode, n_inf)
        res[mode] = (tr, ev)
    for mode in ('inference', 'all'):
        (o0, l0, g0), e0 = res['0']
        (o1, l1, g1), e1 = res[mode]
        assert l0 == l1 and torch.equal(g0, g1) and all(torch.equal(a, b_) for a, b_ in zip(o0, o1)), mode
        assert all(torch.equal(a, b_) for a, b_ in zip(e0, e1)), mode


def test_backward_reports_gradient_buckets_when_final(dev):
    """HfttEngine.backward(on_ready=...) (the hook hftt_hip/ddp.py overlaps its all-reduce on): the three flat ranges tile
    the gradient buffer, and each range already holds its final value at the moment it is reported (stream-ordered
    snapshot), with dropout on so every backward launch is in the plan."""
    from hftt_hip.trainer import TrainStep
    cfg, B = MINI, 2
    model = util.build_model(cfg, 5, dropout=0.1).to(dev)
    model.train()
    ts = TrainStep(model, lr=1e-3)
    eng = ts.engine
    x = (O.synth_spec(B, cfg, salt=3) * 0.5).to(dev)
    ld = _to_dev(O.synth_labels(B, cfg, salt=4), dev)
    ts.forward_backward(x, *ld)                      # builds the plans
    eng.forward(x, training=True)
    eng.loss(B, ld, 1.0, 1.0, with_grad=True)
    eng.flat_grads.fill_(float('nan'))
    snaps = []
    eng.backward(B, on_ready=lambda lo, hi: snaps.append((lo, hi, eng.flat_grads[lo:hi].clone())))
    torch.cuda.synchronize()
    assert [s[0] for s in snaps] == sorted((s[0] for s in snaps), reverse=True)        # time, freq decoder, encoder
    pos = 0
    final = eng.flat_grads.clone()
    for lo, hi, snap in sorted(snaps, key=lambda s: s[0]):
        assert lo == pos
        pos = hi
        for (name, _, o, n) in eng._bound:           # alignment padding between parameters is never written
            if lo <= o < hi:
                assert o + n <= hi
                assert torch.isfinite(snap[o - lo:o - lo + n]).all(), name
                assert torch.equal(snap[o - lo:o - lo + n], final[o:o + n]), name
    assert pos == eng.flat_grads.numel() and len(snaps) == 3


def test_pickle_roundtrip_and_amt_transcript(dev, tmp_path):
    """m_training.py:372-392 (pickle.dump(model), torch.save) and amt.py:21-27,66-118 (pickle.load -> .to -> .eval -> transcript)."""
    from model.amt import AMT
    cfg = MINI
    model = util.build_model(cfg, 5).to(dev)
    x = O.synth_spec(2, cfg, salt=3) * 0.5
    model.eval()
    with torch.no_grad():
        ref = [t.cpu() for t in model(x.to(dev))]
    f = tmp_path / 'model.pkl'
    with open(f, 'wb') as fh:
        pickle.dump(model, fh, protocol=4)
    assert os.path.getsize(f) < 4 * sum(p.numel() for p in model.parameters()) * 1.5 + 200000     # no flat-buffer duplication
    buf = io.BytesIO()
    torch.save({'model_dict': model.state_dict(), 'model': model}, buf)
    config = {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': cfg.n_bin, 'n_bins': cfg.n_bin, 'fft_bins': 2048, 'window_length': 2048,
                          'log_offset': 1e-8, 'window': 'hann', 'pad_mode': 'constant'},
              'input': {'margin_b': cfg.n_margin, 'margin_f': cfg.n_margin, 'num_frame': cfg.n_frame, 'min_value': -18.420681},
              'midi': {'note_min': 21, 'note_max': 21 + cfg.n_note - 1, 'num_note': cfg.n_note, 'num_velocity': cfg.n_velocity}}
    amt = AMT(config, str(f), batch_size=4)
    with torch.no_grad():
        again = [t.cpu() for t in amt.model(x.to(dev))]
    for a, b in zip(again, ref):
        assert max_err(a, b) == 0.0
    n = 3 * cfg.n_frame + 5
    feat = (O.synth_spec(1, O.HfttConfig(n_bin=cfg.n_bin, n_frame=n, n_margin=0), salt=8)[0].T * 0.5).numpy()     # [n, n_bin]
    sd = util.sd_cpu(amt.model)
    outs = amt.transcript(feat)
    ref_outs = O.transcript(feat, lambda s: O.model_forward(sd, s, cfg), cfg, min_value=-18.420681)
    for k, (a, b) in enumerate(zip(outs, ref_outs)):
        assert a.shape == b.shape and a.dtype == b.dtype
        if k % 4 == 3:
            assert (a != b).mean() < 0.01          # argmax of near-ties may flip
        else:
            assert np.abs(a - b).max() < TOL_OUT
    outs_s = amt.transcript_stride(feat, 2)
    ref_s = O.transcript_stride(feat, 2, lambda s: O.model_forward(sd, s, cfg), cfg, min_value=-18.420681)
    for k, (a, b) in enumerate(zip(outs_s, ref_s)):
        if k % 4 != 3:
            assert np.abs(a - b).max() < TOL_OUT
    # state_dict round trip through load_state_dict keeps the engine binding valid
    sd2 = {k: v.clone() + 0.01 for k, v in model.state_dict().items()}
    model.load_state_dict(sd2)
    with torch.no_grad():
        changed = model(x.to(dev))
    assert max_err(changed[0].cpu(), ref[0]) > 0


def test_cpu_call_fails_loudly():
    from hftt_hip import HfttError
    model = util.build_model(MINI, 1)
    with pytest.raises(HfttError):
        model(torch.zeros(1, MINI.n_bin, MINI.n_frame + 2 * MINI.n_margin))


def test_training_train_mirror_fast_and_compat_paths(dev, tmp_path, monkeypatch):
    """training.train.train()/valid() (signature of training/train.py:63,168): torch.optim.Adam + nn criteria (what
    m_training.py passes) and the fused path give the same parameters after two steps (dropout 0)."""
    import torch.nn as nn
    from training import train as T
    from hftt_hip.trainer import FusedAdam
    cfg, B = MINI, 2
    batches = []
    for i in range(2):
        x = O.synth_spec(B, cfg, salt=40 + i) * 0.5
        lo, lf, lm, lv = O.synth_labels(B, cfg, salt=50 + i)
        batches.append((x, lo, lf, lm, lv))
    crits = [nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss(), nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss()]
    res = {}
    for mode in ('compat', 'fast'):
        model = util.build_model(cfg, 21).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3) if mode == 'compat' else FusedAdam(model, lr=1e-3)
        l1 = T.train(model, batches, opt, *crits, 1.0, 1.0, dev, False)
        lv_, n = T.valid(model, batches, *crits, 1.0, 1.0, dev, False)
        assert n == 2 and np.isfinite(l1) and np.isfinite(lv_)
        res[mode] = (l1, lv_, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
    assert abs(res['compat'][0] - res['fast'][0]) < 1e-4
    assert abs(res['compat'][1] - res['fast'][1]) < 2e-3
    # Adam's first steps move every element by ~lr regardless of gradient size: noise-level gradients may take opposite signs
    assert (res['compat'][2] - res['fast'][2]).abs().max().item() <= 4.1e-3
    assert (res['compat'][2] - res['fast'][2]).abs().mean().item() < 2e-5
    # metrics=True (the reference's default last step, m_training.py:466-470): same loss, plus the scores file in the working directory
    # (round 5; the replay test checks the values)
    monkeypatch.chdir(tmp_path)
    lm, nm = T.valid(model, batches, *crits, 1.0, 1.0, dev, True)
    assert nm == 2 and abs(lm - res['fast'][1]) < 1e-6 * abs(lm) and (tmp_path / 'test_performance.json').exists()


def test_inference_wave_to_midi_end_to_end(dev, tmp_path):
    """BASELINE config 5 in miniature: synthetic 16 kHz audio -> HIP log-mel -> clip windows batched through the model ->
    posteriorgrams (vs the oracle on the same features) -> mpe2note -> MIDI file."""
    import json as _json
    from model.amt import AMT
    cfg = O.TINY
    model = util.build_model(cfg, 9)
    util.perturb(model, 10)
    f = tmp_path / 'model.pkl'
    with open(f, 'wb') as fh:
        pickle.dump(model, fh, protocol=4)
    config = _json.loads('{"feature": {"sr": 16000, "hop_sample": 256, "mel_bins": 256, "n_bins": 256, "fft_bins": 2048, "window_length": 2048,'
                         ' "log_offset": 1e-8, "window": "hann", "pad_mode": "constant"}, "input": {"margin_b": 32, "margin_f": 32, "num_frame": 128,'
                         ' "min_value": -18.420681}, "midi": {"note_min": 21, "note_max": 108, "num_note": 88, "num_velocity": 128}}')
    amt = AMT(config, str(f), batch_size=8)
    sr, dur = 16000, 2.5
    t = torch.arange(int(sr * dur)) / sr
    wave = sum(0.2 * torch.sin(2 * np.pi * f0 * t) * torch.exp(-3.0 * (t - t0).clamp(min=0)) * (t >= t0)
               for f0, t0 in ((196.0, 0.1), (246.9, 0.6), (329.6, 1.2), (440.0, 1.7)))
    feat = amt.wave2feature(wave.unsqueeze(0), sr)
    assert feat.shape == (1 + len(t) // 256, 256)
    assert max_err(feat, O.logmel_dft(wave)) < 5e-3      # log domain; near-silent bins carry fp32 FFT noise
    outs = amt.transcript(feat.numpy())
    sd = util.sd_cpu(amt.model)
    ref = O.transcript(feat.numpy(), lambda s: O.model_forward(sd, s, cfg), cfg, min_value=-18.420681)
    for k, (a, b) in enumerate(zip(outs, ref)):
        assert a.shape == b.shape == (256, 88)
        if k % 4 != 3:
            assert np.abs(a - b).max() < TOL_OUT, k
    notes = amt.mpe2note(a_onset=outs[4], a_offset=outs[5], a_mpe=outs[6], a_velocity=outs[7], thred_onset=0.3, thred_offset=0.3, thred_mpe=0.3)
    notes_ref = amt.mpe2note(a_onset=ref[4], a_offset=ref[5], a_mpe=ref[6], a_velocity=ref[7], thred_onset=0.3, thred_offset=0.3, thred_mpe=0.3)
    assert abs(len(notes) - len(notes_ref)) <= max(2, len(notes_ref) // 50)      # untrained weights: only near-threshold frames may differ
    mid = tmp_path / 'out.mid'
    amt.note2midi(notes, str(mid))
    assert mid.read_bytes()[:4] == b'MThd'


def test_frozen_weights_prepare_once_and_follow_explicit_changes(dev):
    """model.hftt_freeze_weights(): eval forwards skip the per-call operand preparation; results are unchanged; load_state_dict, train() and
    a FusedAdam step end / refresh the promise (the reference reads its weights at every call)."""
    from hftt_hip.trainer import FusedAdam, TrainStep
    cfg = MINI
    model = util.build_model(cfg, 9).to(dev)
    model.hftt_precision = 'bf16'
    x = (O.synth_spec(2, cfg, salt=4) * 0.5).to(dev)
    model.eval()
    with torch.no_grad():
        ref = [t.clone() for t in model(x)]
        model.hftt_freeze_weights(True)
        eng = model.hftt_engine()
        calls = []
        orig = eng.lib.hftt_prep_weights
        a = [t.clone() for t in model(x)]
        assert eng.frozen_weights and eng._prepared_frozen
        b = [t.clone() for t in model(x)]
    for r, u, v in zip(ref, a, b):
        assert torch.equal(r, u) and torch.equal(r, v)
    sd = {k: v.clone() + 0.02 for k, v in model.state_dict().items()}
    model.load_state_dict(sd)                                   # explicit change: prepared again
    with torch.no_grad():
        c = model(x)
    assert not torch.equal(c[0], ref[0])
    model.train()                                               # training: never frozen
    assert not model.hftt_engine().frozen_weights
    ts = TrainStep(model, optimizer=FusedAdam(model.parameters(), lr=1e-2))
    labels = [t.to(dev).contiguous() for t in O.synth_labels(2, cfg, salt=5)]
    ts(x, *labels)
    model.eval()
    with torch.no_grad():
        d = [t.clone() for t in model(x)]                        # sees the updated parameters
        model.hftt_freeze_weights(False)
        e = model(x)
    assert not torch.equal(d[0], c[0])
    for u, v in zip(d, e):
        assert torch.equal(u, v)


@pytest.mark.parametrize('precision', ['x3', 'parity'])
def test_gradient_accumulation_over_two_backwards(dev, precision):
    """Two loss.backward() calls WITHOUT zeroing in between (the autograd / compatibility path): p.grad must hold g1 + g2.  After the first
    backward p.grad is a view of the engine's flat gradient buffer, which the second backward overwrites before autograd accumulates --
    hftt_hip/autograd.py puts the held gradients back and hands the new ones over as a copy."""
    cfg, B = MINI, 2
    model = util.build_model(cfg, 31)
    util.perturb(model, 32)
    sd = util.sd_cpu(model)
    x1 = O.synth_spec(B, cfg, salt=3) * 0.5; x2 = O.synth_spec(B, cfg, salt=4) * 0.5
    l1 = O.synth_labels(B, cfg, salt=5); l2 = O.synth_labels(B, cfg, salt=6)
    _, _, g1 = _oracle_run(sd, cfg, x1, l1)
    _, _, g2 = _oracle_run(sd, cfg, x2, l2)
    model = model.to(dev)
    model.hftt_precision = precision
    model.train()
    for x, l in ((x1, l1), (x2, l2)):
        O.spec2midi_loss(model(x.to(dev)), *_to_dev(l, dev)).backward()
    tol = 3e-3 if precision == 'x3' else 2e-3
    for name, p in model.named_parameters():
        ref = g1[name] + g2[name]
        scale = max(ref.abs().max().item(), 1e-6)
        if ref.abs().max().item() < 1e-7:
            continue
        assert max_err(p.grad, ref) / scale < tol, (name, max_err(p.grad, ref) / scale)
        single = max_err(p.grad, g2[name] * 2) / scale          # what the aliasing bug produced
        assert single > 10 * tol or max_err(g1[name], g2[name]) / scale < 20 * tol, name
    # zero_grad(set_to_none=False) keeps the views; a further backward then yields exactly that pass's gradients
    model.zero_grad(set_to_none=False)
    O.spec2midi_loss(model(x1.to(dev)), *_to_dev(l1, dev)).backward()
    for name, p in model.named_parameters():
        scale = max(g1[name].abs().max().item(), 1e-6)
        if g1[name].abs().max().item() >= 1e-7:
            assert max_err(p.grad, g1[name]) / scale < tol, name
