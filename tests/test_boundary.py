"""Drop-in boundary without a GPU: class paths, constructor signatures, parameter names / shapes,
initialisation stream, pickling, and the refusal to compute on the CPU."""
import inspect
import io
import pickle

import numpy as np
import pytest
import torch

import util
from util import O


def test_constructor_signatures_match_the_reference():
    from model import model_spec2midi as M
    sig = lambda f: list(inspect.signature(f).parameters)[1:]
    assert sig(M.Encoder_SPEC2MIDI.__init__) == ['n_margin', 'n_frame', 'n_bin', 'cnn_channel', 'cnn_kernel', 'hid_dim', 'n_layers',
                                                 'n_heads', 'pf_dim', 'dropout', 'device']              # model_spec2midi.py:42
    assert sig(M.Decoder_SPEC2MIDI.__init__) == ['n_frame', 'n_bin', 'n_note', 'n_velocity', 'hid_dim', 'n_layers', 'n_heads', 'pf_dim',
                                                 'dropout', 'device']                                    # :113
    assert sig(M.Model_SPEC2MIDI.__init__) == ['encoder', 'decoder']                                     # :10
    assert sig(M.EncoderLayer.__init__) == ['hid_dim', 'n_heads', 'pf_dim', 'dropout', 'device']
    assert sig(M.MultiHeadAttentionLayer.__init__) == ['hid_dim', 'n_heads', 'dropout', 'device']
    assert sig(M.PositionwiseFeedforwardLayer.__init__) == ['hid_dim', 'pf_dim', 'dropout']
    from model.amt import AMT
    assert sig(AMT.__init__)[:4] == ['config', 'model_path', 'batch_size', 'verbose_flag']       # the reference's four, positional order kept
    assert sig(AMT.__init__)[4:] == ['rank', 'world', 'device', 'gather']                           # keyword additions of the multi-GPU scatter (defaults: None, None, None, 'host')
    assert sig(AMT.transcript) == ['a_feature', 'mode', 'ablation_flag']
    assert sig(AMT.transcript_stride) == ['a_feature', 'n_offset', 'mode', 'ablation_flag']
    assert sig(AMT.mpe2note) == ['a_onset', 'a_offset', 'a_mpe', 'a_velocity', 'thred_onset', 'thred_offset', 'thred_mpe',
                                 'mode_velocity', 'mode_offset']
    with pytest.raises(AssertionError):
        M.MultiHeadAttentionLayer(10, 3, 0.0, 'cpu')                                                     # :311


@pytest.mark.parametrize('name,n_keys,n_params', [('tiny_b2', 115, 279646), ('paper_b1', 165, 5516574)])
def test_state_dict_names_shapes_and_init_stream(name, n_keys, n_params):
    g = util.golden(name)
    cfg = util.cfg_from_golden(g)
    model = util.build_model(cfg, int(g['seed']))
    sd = model.state_dict()
    ref_names = [k[len('sdsum.'):] for k in g.files if k.startswith('sdsum.')]
    assert list(sd.keys()) == ref_names                        # same names, same order as the reference's state_dict
    assert len(sd) == n_keys and sum(p.numel() for p in model.parameters()) == n_params
    assert model.encoder_spec2midi.layers_freq[0].self_attention.fc_q.weight.shape == (cfg.hid_dim, cfg.hid_dim)
    assert model.decoder_spec2midi.fc_velocity_time.weight.shape == (cfg.n_velocity, cfg.hid_dim)
    # apply(initialize_weights) touched exactly the >1-dim weights: their checksums equal the reference's
    for k, v in sd.items():
        if v.dim() > 1:
            assert abs(v.double().sum().item() - g['sdsum.' + k][0]) < 1e-9 * max(1.0, g['sdsum.' + k][1]), k
    # load_state_dict of an oracle-side dict works, with strict key matching
    model.load_state_dict({k: torch.zeros_like(v) for k, v in sd.items()})
    assert all((p == 0).all() for p in model.parameters())


def test_pickle_and_torch_save_roundtrip_on_cpu():
    model = util.build_model(util.MINI, 3, dropout=0.1)
    blob = pickle.dumps(model, protocol=4)
    import pickletools
    globs = {arg for op, arg, _ in pickletools.genops(blob) if op.name in ('GLOBAL', 'STACK_GLOBAL') and arg}
    m2 = pickle.loads(blob)
    assert type(m2).__module__ == 'model.model_spec2midi' and type(m2).__name__ == 'Model_SPEC2MIDI'
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert m2.encoder_spec2midi.dropout.p == 0.1
    buf = io.BytesIO()
    torch.save({'model_dict': model.state_dict(), 'model': model}, buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    assert torch.equal(ck['model'].decoder_spec2midi.fc_mpe_time.weight, model.decoder_spec2midi.fc_mpe_time.weight)


def test_no_cpu_fallback():
    from hftt_hip import HfttError
    from hftt_hip import ops
    model = util.build_model(util.MINI, 1)
    with pytest.raises(HfttError, match='no CPU fallback'):
        model(torch.zeros(1, util.MINI.n_bin, util.MINI.n_frame + 2 * util.MINI.n_margin))
    with pytest.raises(HfttError):
        model.encoder_spec2midi(torch.zeros(1, 48, 24))
    with pytest.raises(HfttError):
        ops.gemm_nt(torch.zeros(4, 32), torch.zeros(8, 32))
    import os
    src = open(os.path.join(util.ROOT, 'nylon-amt_amd', 'hftt_hip', 'engine.py')).read() + open(os.path.join(util.ROOT, 'nylon-amt_amd', 'model', 'model_spec2midi.py')).read()
    assert 'oracle' not in src                      # the product path never touches the oracle


def test_only_tests_smoke_and_the_cpu_baseline_reach_the_oracle():
    """oracle/ is test infrastructure: no file of the package or of tools/ imports it, directly or through tests/util (which holds it as `O`)."""
    import os
    import re
    pat = re.compile(r'^\s*(from oracle|import oracle|import util\b|from util import|import test_)', re.M)
    offenders = []
    for top in ('nylon-amt_amd', 'tools'):
        for dp, dn, fn in os.walk(os.path.join(util.ROOT, top)):
            dn[:] = [d for d in dn if d not in ('__pycache__', 'build', 'lib')]
            for f in fn:
                if f.endswith(('.py', '.sh')):
                    src = open(os.path.join(dp, f), errors='replace').read()
                    if pat.search(src) or re.search(r'oracle[/.]hftt_oracle', src):
                        offenders.append(os.path.relpath(os.path.join(dp, f), util.ROOT))
    assert offenders == [], offenders


def test_bench_measured_leg_is_oracle_free_and_configs_match():
    """bench.py may use oracle/ only for its reported cpu_baseline; its own workload tables must equal the oracle's configurations
    (both restate m_training.py's defaults), and its model builder must produce the reference's parameter set."""
    import os
    import re
    import bench
    from oracle import hftt_oracle as O
    for name, ocfg in (('paper', O.PAPER), ('tiny', O.TINY)):
        assert bench.CONFIGS[name]._asdict() == {k: getattr(ocfg, k) for k in bench.BenchCfg._fields}, name
    src = open(os.path.join(util.ROOT, 'bench.py')).read()
    imports = [m.start() for m in re.finditer(r'^\s*(from oracle|import oracle)', src, re.M)]
    a, b = src.index('def cpu_baseline('), src.index('def pmc_traffic_bytes(')
    assert len(imports) == 1 and a < imports[0] < b            # the single import sits inside cpu_baseline()
    assert 'tests' not in src[src.index('for p in (ROOT'):src.index('import collections')]     # tests/ is not on bench's path
    m1 = bench.build_model(bench.CONFIGS['tiny'], 5, 0.1, 'cpu')
    m2 = util.build_model(O.TINY, 5, dropout=0.1)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    assert list(sd1.keys()) == list(sd2.keys())
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)        # same construction order and init stream as m_training.py
    x, lab = bench.synthetic_batch(bench.CONFIGS['tiny'], 2, 1, 'cpu')
    assert x.shape == (2, 256, 192) and x.min() >= -18.420681 - 1e-6 and x.max() <= 6.0
    assert [t.shape for t in lab] == [(2, 128, 88)] * 4 and lab[3].dtype == torch.int64 and int(lab[3].max()) < 128


@pytest.mark.parametrize('p', [0.1, 0.25, 0.5, 0.3])
def test_dropout_is_unbiased(p):
    """nn.Dropout's contract (model_spec2midi.py:95, 348, 376): E[dropout(x)] = x.  The device generator keeps an element with probability
    thr / 256 (8-bit threshold) and scales the kept ones by 256 / thr (csrc/hftt_common.h: hftt_keep_scale), NOT by 1 / (1 - p): the mean
    of keep * scale over many elements is 1 to sampling noise for every p (with 1 / (1 - p) it was 0.9983 at p = 0.1)."""
    import numpy as np
    n = 1 << 22
    keep = util.keep_mask(0x1234ABCD, 7, np.arange(n, dtype=np.uint64), p)
    sc = util.keep_scale(p)
    mean = float(keep.mean()) * sc
    assert abs(mean - 1.0) < 1e-3, (p, mean)
    thr = round(256.0 / sc)
    assert abs(sc * thr - 256.0) < 1e-3 and abs(thr / 256.0 - (1.0 - p)) <= 0.5 / 256 + 1e-9
    from hftt_hip.engine import keep_scale
    assert keep_scale(p) == sc and keep_scale(0.0) == 1.0
