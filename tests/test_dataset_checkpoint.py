"""Clip contract (training/dataset.py MyDataset), dataset assembly (corpus/make_dataset.py) and reference-format checkpoints
(m_training.py:268-299, 372-392; amt.py:21-27) against fixtures generated FROM THE REFERENCE by tests/golden/make_golden_r2.py."""
import importlib.util
import os
import pickle

import numpy as np
import pytest
import torch

import util

G = util.GOLDEN
MICRO_DS = {'feature': {'mel_bins': 16, 'n_bins': 16, 'log_offset': 1e-8, 'sr': 16000, 'hop_sample': 256},
            'input': {'margin_b': 4, 'margin_f': 4, 'num_frame': 8, 'min_value': -18.420681, 'max_value': 0.0},
            'midi': {'note_min': 21, 'num_note': 6, 'num_velocity': 8}}


def _mods():
    from training.dataset import MyDataset, DeviceClipStore
    from corpus.make_dataset import synth_store
    return MyDataset, DeviceClipStore, synth_store


def _store_files(tmp_path, store):
    paths = {}
    for k, v in store.items():
        paths[k] = str(tmp_path / (k + '.pkl'))
        with open(paths[k], 'wb') as f:
            pickle.dump(v, f, protocol=4)
    return paths


def test_dataset_assembly_reproduces_the_fixture_store():
    """corpus.make_dataset.synth_store is deterministic: the store the golden generator fed to the reference is the one built here."""
    _, _, synth_store = _mods()
    g = np.load(os.path.join(G, 'dataset.npz'))
    store = synth_store(MICRO_DS, [20, 13], seed=1234)
    for k, v in store.items():
        assert v.dtype == g['store.' + k].dtype and np.array_equal(v, g['store.' + k]), k
    cin = MICRO_DS['input']
    gap = cin['margin_f'] + cin['num_frame'] - 1
    assert store['feature'].shape[0] == cin['margin_b'] + (20 + gap) + (13 + gap)                 # make_dataset.py:34,55
    assert np.array_equal(store['idx'], np.concatenate([np.arange(4, 24), np.arange(24 + gap, 24 + gap + 13)]).astype(np.int32))
    assert np.all(store['feature'][:4] == np.float32(np.log(1e-8)))                                 # padding value :105-113


@pytest.mark.parametrize('tag', ['raw', 'scaled'])
def test_assemble_store_equals_the_reference_make_dataset(tag):
    """corpus.make_dataset.assemble_store against the arrays the REFERENCE's make_dataset (corpus/make_dataset.py:11-239) wrote for the same
    per-file features / labels (tests/golden/make_golden_r3.py ran it): concatenation, padding, idx, a label track longer / shorter than
    its feature array, and both feature branches (:100-116: raw + log(log_offset) padding, or scaled by max_value + zero padding); plus the
    two config keys its __main__ writes back (:274-278, :305-306)."""
    import copy
    from corpus.make_dataset import assemble_store, prepare_config, finalize_config
    g = np.load(os.path.join(G, 'store.npz'))
    cfg = {'feature': {'mel_bins': 16, 'log_offset': 1e-8, 'sr': 16000, 'hop_sample': 256},
           'input': {'margin_b': 4, 'margin_f': 4, 'num_frame': 8},
           'midi': {'note_min': 21, 'num_note': 6, 'num_velocity': 8}}
    cfg = prepare_config(copy.deepcopy(cfg), float(g[tag + '.max_value']))
    n_files = len([k for k in g.files if k.startswith('in.') and k.endswith('.feature')])
    feats = [g['in.%d.feature' % i] for i in range(n_files)]
    labels = [{k: g['in.%d.%s' % (i, k)] for k in ('onset', 'offset', 'mpe', 'velocity')} for i in range(n_files)]
    store = assemble_store(feats, labels, cfg)
    for mine, theirs in (('idx', 'idx'), ('feature', 'feature'), ('label_mpe', 'label_mpe'), ('label_onset', 'label_onset'),
                         ('label_offset', 'label_offset'), ('label_velocity', 'label_velocity')):
        ref = g['%s.%s' % (tag, theirs)]
        assert store[mine].shape == ref.shape and store[mine].dtype == ref.dtype, (mine, store[mine].dtype, ref.dtype)
        assert np.array_equal(store[mine], ref), mine
    finalize_config(cfg)
    assert isinstance(cfg['input']['min_value'], float) and cfg['input']['min_value'] == float(g[tag + '.min_value'])
    assert cfg['feature']['n_bins'] == int(g[tag + '.n_bins']) == cfg['feature']['mel_bins']
    assert cfg['input']['max_value'] == float(g[tag + '.max_value'])


@pytest.mark.parametrize('n_slice', [1, 4])
def test_mydataset_equals_the_reference_class(tmp_path, n_slice):
    MyDataset, _, synth_store = _mods()
    g = np.load(os.path.join(G, 'dataset.npz'))
    store = {k[6:]: g[k] for k in g.files if k.startswith('store.')}
    p = _store_files(tmp_path, store)
    ds = MyDataset(p['feature'], p['label_onset'], p['label_offset'], p['label_mpe'], p['label_velocity'], p['idx'], MICRO_DS, n_slice)
    assert len(ds) == int(g['n%d.len' % n_slice])
    assert np.array_equal(ds.idx.numpy(), g['n%d.idx' % n_slice])
    for tag, i in (('first', 0), ('second', 1), ('last', len(ds) - 1)):
        item = ds[i]
        assert len(item) == 5
        for name, t in zip(('spec', 'onset', 'offset', 'mpe', 'velocity'), item):
            ref = g['n%d.%s.%s' % (n_slice, tag, name)]
            assert str(t.dtype) == str(g['n%d.%s.%s.dtype' % (n_slice, tag, name)]), (name, t.dtype)
            assert tuple(t.shape) == ref.shape and np.array_equal(t.contiguous().numpy(), ref), (tag, name)
    # collated batches (what train() receives) == the device-resident store's gather (CPU device here)
    _, DeviceClipStore, _ = _mods()
    dcs = DeviceClipStore(ds, 'cpu')
    ids = list(range(0, len(ds), 2))[:4]
    ref = torch.utils.data.default_collate([ds[i] for i in ids])
    for a, b in zip(dcs.batch(ids), ref):
        assert a.dtype == b.dtype and torch.equal(a, b)
    chunks = list(dcs.loader(3, rank=1, world=2))
    assert sum(c[0].shape[0] for c in chunks) == len(range(1, len(ds), 2))
    # the clip contract at the real constants (config 1 of SURVEY 8(d)): shapes and dtypes train.py:72-76 / :106-132 rely on
    real = {'feature': {'mel_bins': 256, 'n_bins': 256, 'log_offset': 1e-8}, 'input': {'margin_b': 32, 'margin_f': 32, 'num_frame': 128},
            'midi': {'num_note': 88, 'num_velocity': 128}}
    big = synth_store(real, [300, 180], seed=1234)
    dsb = MyDataset.from_arrays(big['feature'], big['label_onset'], big['label_offset'], big['label_mpe'], big['label_velocity'], big['idx'], real, 100)
    assert len(dsb) == 4
    spec, lo, lf, lm, lv = dsb[3]
    assert tuple(spec.shape) == (256, 192) and spec.dtype == torch.float32 and not spec.is_contiguous()
    assert tuple(lo.shape) == (128, 88) and lm.dtype == torch.float32 and lv.dtype == torch.int64


def _ckpt_cfg():
    g = np.load(os.path.join(G, 'ref_ckpt.npz'))
    return g, {str(k): int(v) for k, v in zip(g['cfg_keys'], g['cfg'])}


def test_reference_pickle_loads_into_this_model():
    """pickle.load of a model written by the REFERENCE resolves to model.model_spec2midi of this repo (amt.py:24-25)."""
    import model.model_spec2midi as M
    g, cfg = _ckpt_cfg()
    with open(os.path.join(G, 'ref_ckpt_model.pkl'), 'rb') as f:
        model = pickle.load(f)
    assert type(model) is M.Model_SPEC2MIDI and type(model.encoder_spec2midi.layers_freq[0]) is M.EncoderLayer
    assert model.hftt_config() == cfg
    assert model.hftt_precision in ('x3', 'parity', 'bf16') and model.training is False
    sd = model.state_dict()
    keys = [k[3:] for k in g.files if k.startswith('sd.')]
    assert sorted(sd.keys()) == sorted(keys)
    for k in keys:
        assert np.array_equal(sd[k].numpy(), g['sd.' + k]), k


def test_reference_dat_checkpoint_resumes_optimizer_and_scheduler():
    """m_training.py:268-275: model_dict / optimizer_dict / scheduler_dict of a reference .dat load into this repo's model, FusedAdam and
    ReduceLROnPlateau; saving again gives dictionaries of the same layout."""
    from hftt_hip.trainer import FusedAdam
    g, cfg = _ckpt_cfg()
    ck = torch.load(os.path.join(G, 'ref_ckpt_model.dat'), map_location='cpu', weights_only=False)
    assert {'optimizer_dict', 'scheduler_dict', 'model_dict', 'model', 'epoch', 'div', 'best_loss_valid'} <= set(ck)
    c = util.O.HfttConfig(**cfg)
    model = util.build_model(c, 1)
    model.load_state_dict(ck['model_dict'])
    for k, v in model.state_dict().items():
        assert np.array_equal(v.numpy(), g['sd.' + k]), k
    opt = FusedAdam(model.parameters(), lr=3e-4)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt)                 # m_training.py:147 -- needs a real torch Optimizer
    opt.load_state_dict(ck['optimizer_dict'])
    sched.load_state_dict(ck['scheduler_dict'])
    assert opt.param_groups[0]['lr'] == ck['optimizer_dict']['param_groups'][0]['lr'] == 1e-4
    assert opt.step_count == 1
    out = opt.state_dict()
    ref = ck['optimizer_dict']
    assert set(out['state'].keys()) == set(ref['state'].keys()) and out['param_groups'][0]['params'] == ref['param_groups'][0]['params']
    for i in ref['state']:
        for key in ('step', 'exp_avg', 'exp_avg_sq'):
            assert torch.equal(torch.as_tensor(out['state'][i][key]).float(), torch.as_tensor(ref['state'][i][key]).float()), (i, key)
    assert sched.state_dict()['best'] == ck['scheduler_dict']['best']


def test_fused_adam_is_a_torch_optimizer_and_drives_reduce_lr_on_plateau():
    from hftt_hip.trainer import FusedAdam
    from hftt_hip._capi import HfttError
    model = util.build_model(util.MINI, 3)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=1)
    for loss in (1.0, 1.0, 1.0, 1.0):
        sched.step(loss)
    assert opt.param_groups[0]['lr'] == pytest.approx(1e-4)               # reduced by the default factor 0.1
    sd = opt.state_dict()
    ref = torch.optim.Adam(model.parameters(), lr=1e-3).state_dict()
    assert set(ref['param_groups'][0].keys()) <= set(sd['param_groups'][0].keys())
    assert sd['state'] == {} and 'scheduler' not in sd
    opt.zero_grad()
    with pytest.raises(HfttError):
        opt.step()                                                        # parameters on the CPU: no engine, no fallback


# ---------------------------------------------------------------------------------------------------------------------------------
# f4: label generation (corpus/conv_note2label.py) against the reference function's output
@pytest.mark.parametrize('case', [0, 1])
@pytest.mark.parametrize('flag', [False, True])
def test_note2label_equals_the_reference(case, flag):
    from corpus.conv_note2label import note2label, note2label_arrays
    from corpus.make_dataset import assemble_store
    g = np.load(os.path.join(G, 'labels.npz'))
    notes = [{'pitch': int(r[0]), 'onset': float(r[1]), 'offset': float(r[2]), 'velocity': int(r[3])} for r in g['c%d.notes' % case]]
    config = {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': 8, 'log_offset': 1e-8}, 'midi': {'note_min': 21, 'num_note': 88, 'num_velocity': 128},
              'input': {'margin_b': 32, 'margin_f': 32, 'num_frame': 128}}
    lab = note2label_arrays(config, notes, flag)
    for k in ('mpe', 'onset', 'offset', 'velocity'):
        ref = g['c%d.%d.%s' % (case, int(flag), k)]
        assert lab[k].dtype == ref.dtype and lab[k].shape == ref.shape, k
        assert np.array_equal(lab[k], ref), (k, int((lab[k] != ref).sum()))
    as_lists = note2label(config, notes, flag)                       # the reference's return type: nested lists
    assert set(as_lists) == {'mpe', 'onset', 'offset', 'velocity'} and isinstance(as_lists['mpe'][0][0], bool)
    assert as_lists['onset'][5] == lab['onset'][5].tolist()
    # properties of the targets: peaks below one between frames, re-struck pitch without an offset target, mpe covers onset..offset
    assert 0.5 < lab['onset'].max() <= 1.0 and lab['velocity'].max() <= 127 and lab['velocity'].min() >= 0
    p60 = 60 - 21
    f_restrike = int(0.5 * 62.5 + 0.5)
    assert lab['offset'][f_restrike, p60] == 0.0 and lab['onset'][f_restrike, p60] > 0.5
    assert lab['mpe'][:f_restrike + 1, p60].all()
    # ... and they feed the store assembly (labels longer / shorter than the feature of a file are both handled, make_dataset.py:52)
    feat = np.zeros((lab['mpe'].shape[0] - 7, 8), np.float32)
    store = assemble_store([feat], [lab], config)
    assert store['idx'].shape[0] == lab['mpe'].shape[0] and store['label_velocity'].dtype == np.int8
    assert np.array_equal(store['label_onset'][32:32 + lab['onset'].shape[0]], lab['onset'])


def test_note2label_edge_cases():
    from corpus.conv_note2label import note2label_arrays
    config = {'feature': {'sr': 16000, 'hop_sample': 256}, 'midi': {'note_min': 21, 'num_note': 88}}
    empty = note2label_arrays(config, [], False)
    assert empty['mpe'].shape == (1, 88) and not empty['mpe'].any()
    one = note2label_arrays(config, [{'pitch': 21, 'onset': 0.0, 'offset': 0.0, 'velocity': 5}], True)      # zero-length note at t = 0
    assert one['mpe'].shape == (1, 88) and one['mpe'][0, 0] and one['onset'][0, 0] == 1.0 and one['offset'][0, 0] == 0.0 and one['velocity'][0, 0] == 5
