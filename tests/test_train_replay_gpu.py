"""The reference's training driver replayed against this repo's mirrors on the GPU: training/m_training.py (3)-(7) -- construct ->
.to(device) -> apply(initialize_weights) -> Adam(model.parameters()) -> ReduceLROnPlateau -> MyDataset / DataLoader -> train() -> valid()
-> pickle.dump / torch.save -> scheduler.step -> resume (load_state_dict x3) -- with ONE line changed (optim.Adam -> FusedAdam), and the
same driver unchanged (torch.optim.Adam: the compatibility path).  Plus: reference-made checkpoints on the GPU (tests/golden/ref_ckpt*),
and the data-parallel step as two real processes (gloo) on one GPU."""
import os
import pickle

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.multiprocessing as mp

import util
from util import O, MINI, max_err

pytestmark = pytest.mark.gpu
G = util.GOLDEN


def _ds_config(cfg):
    return {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': cfg.n_bin, 'n_bins': cfg.n_bin, 'log_offset': 1e-8},
            'input': {'margin_b': cfg.n_margin, 'margin_f': cfg.n_margin, 'num_frame': cfg.n_frame, 'min_value': -18.420681, 'max_value': 0.0},
            'midi': {'note_min': 21, 'note_max': 21 + cfg.n_note - 1, 'num_note': cfg.n_note, 'num_velocity': cfg.n_velocity}}


def _write_store(d, cfg, frames, seed, split):
    from corpus.make_dataset import synth_store
    store = synth_store(_ds_config(cfg), frames, seed=seed)
    paths = []
    for k in ('feature', 'label_onset', 'label_offset', 'label_mpe', 'label_velocity', 'idx'):
        os.makedirs(os.path.join(d, k), exist_ok=True)
        paths.append(os.path.join(d, k, split + '.pkl'))
        with open(paths[-1], 'wb') as f:
            pickle.dump(store[k], f, protocol=4)
    return paths


class _Driver:
    """m_training.py:109-157 restated as calls (the reference file is a script, not importable as a function)."""

    def __init__(self, dev, d_out, cfg, make_optimizer, seed=1234, dropout=0.1, lr=1e-3, batch=4, n_slice=4):
        from model.model_spec2midi import Encoder_SPEC2MIDI, Decoder_SPEC2MIDI, Model_SPEC2MIDI
        from training import dataset
        torch.manual_seed(seed)                                                                         # :109
        encoder = Encoder_SPEC2MIDI(cfg.n_margin, cfg.n_frame, cfg.n_bin, cfg.cnn_channel, cfg.cnn_kernel, cfg.hid_dim, cfg.enc_layer,
                                    cfg.enc_head, cfg.pf_dim, dropout, dev)                               # :117-127
        decoder = Decoder_SPEC2MIDI(cfg.n_frame, cfg.n_bin, cfg.n_note, cfg.n_velocity, cfg.hid_dim, cfg.dec_layer, cfg.dec_head, cfg.pf_dim,
                                    dropout, dev)                                                        # :128-137
        model = Model_SPEC2MIDI(encoder, decoder)                                                        # :138
        model = model.to(dev)                                                                            # :140
        model.apply(util.initialize_weights)                                                             # :141 (on the device)
        self.model = model
        self.optimizer = make_optimizer(model.parameters(), lr)                                          # :146
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer)                      # :147
        self.crits = [nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss(), nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss()]
        self.dev, self.d_out, self.cfg = dev, d_out, cfg
        conf = _ds_config(cfg)
        tr = _write_store(d_out, cfg, [70, 45], 11, 'train')
        va = _write_store(d_out, cfg, [40], 12, 'valid')
        self.dataset_train = dataset.MyDataset(*tr, conf, n_slice)                                       # :222-229
        self.dataset_valid = dataset.MyDataset(*va, conf, n_slice)
        self.loader_train = torch.utils.data.DataLoader(self.dataset_train, batch_size=batch, shuffle=False)   # (shuffle off: comparable runs)
        self.loader_valid = torch.utils.data.DataLoader(self.dataset_valid, batch_size=batch, shuffle=False)
        self.best = float('inf')

    def epoch(self, epoch):
        from training import train
        lt = train.train(self.model, self.loader_train, self.optimizer, *self.crits, 1.0, 1.0, self.dev, False)      # :322-330
        lv, n = train.valid(self.model, self.loader_valid, *self.crits, 1.0, 1.0, self.dev, metrics=False)           # :362-366
        lv /= n
        base = os.path.join(self.d_out, 'model_%03d_000' % epoch)
        with open(base + '.pkl', 'wb') as f:
            pickle.dump(self.model, f, protocol=4)                                                       # :372-373
        torch.save({'epoch': epoch, 'div': 0, 'epoch_loss_train': lt, 'epoch_loss_valid': lv, 'best_epoch': epoch, 'best_div': 0,
                    'best_loss_valid': self.best, 'optimizer_dict': self.optimizer.state_dict(), 'scheduler_dict': self.scheduler.state_dict(),
                    'model_dict': self.model.state_dict(),
                    'random': {'torch': torch.get_rng_state(), 'torch_random': torch.random.get_rng_state(), 'cuda': torch.cuda.get_rng_state(),
                               'cuda_all': torch.cuda.get_rng_state_all()},
                    'model': self.model}, base + '.dat')                                                 # :374-392
        self.best = min(self.best, lv)
        self.scheduler.step(lv)                                                                          # :437
        return lt, lv

    def test(self, batch=4, n_slice=4):
        """(7-3) `-valid_test` is True by default (m_training.py:64): the last thing an unchanged run does is valid(metrics=True) :466-470"""
        from training import dataset, train
        te = _write_store(self.d_out, self.cfg, [52], 13, 'test')
        dataset_test = dataset.MyDataset(*te, _ds_config(self.cfg), n_slice)                              # :238-246
        loader_test = torch.utils.data.DataLoader(dataset_test, batch_size=batch, shuffle=False)
        loss, n = train.valid(self.model, loader_test, *self.crits, 1.0, 1.0, self.dev, metrics=True)      # :466-470
        return loss / n, loader_test                                                                     # :471

    def resume(self, epoch):
        ck = torch.load(os.path.join(self.d_out, 'model_%03d_000.dat' % epoch), weights_only=False)      # :268
        self.model.load_state_dict(ck['model_dict'])                                                     # :270
        self.optimizer.load_state_dict(ck['optimizer_dict'])                                             # :272
        self.scheduler.load_state_dict(ck['scheduler_dict'])                                             # :273
        return ck

    def params(self):
        return torch.cat([p.detach().reshape(-1) for p in self.model.parameters()]).cpu()


def test_m_training_replay_fused_adam_and_unchanged_driver(dev, tmp_path):
    from hftt_hip.trainer import FusedAdam
    cfg = MINI
    runs = {}
    for kind, mk in (('fused', lambda ps, lr: FusedAdam(ps, lr=lr)), ('torch', lambda ps, lr: torch.optim.Adam(ps, lr=lr))):
        d = _Driver(dev, str(tmp_path / kind), cfg, mk, dropout=0.0)
        os.makedirs(d.d_out, exist_ok=True)
        hist = [d.epoch(e) for e in range(3)]
        runs[kind] = (hist, d.params(), d)
        assert all(np.isfinite(v) for h in hist for v in h)
        assert hist[-1][0] < hist[0][0]                                         # the training loss goes down
    (hf, pf, df), (ht, pt, dt) = runs['fused'], runs['torch']
    # the one-line swap changes nothing but speed: same losses, same parameters (dropout 0; Adam's sign-like first steps bound the tail)
    for a, b in zip(hf, ht):
        assert abs(a[0] - b[0]) < 2e-3 * abs(b[0]) and abs(a[1] - b[1]) < 2e-3 * abs(b[1]), (hf, ht)
    assert (pf - pt).abs().mean().item() < 1e-4
    # optimizer state has torch Adam's layout and values
    sf, st = df.optimizer.state_dict(), dt.optimizer.state_dict()
    assert sf['param_groups'][0]['params'] == st['param_groups'][0]['params'] and set(sf['state']) == set(st['state'])
    n_steps = 3 * len(df.loader_train)
    ma, mb = [], []
    for i in st['state']:
        assert float(sf['state'][i]['step']) == float(st['state'][i]['step']) == n_steps
        a, b = sf['state'][i]['exp_avg'].cpu(), st['state'][i]['exp_avg'].cpu()
        assert a.shape == b.shape
        ma.append(a.reshape(-1)); mb.append(b.reshape(-1))
    ma, mb = torch.cat(ma).double(), torch.cat(mb).double()
    # (per-tensor relative errors mean nothing for tensors whose gradient is rounding noise, e.g. fc_k.bias: compare the whole vector)
    cos = float((ma * mb).sum() / (ma.norm() * mb.norm()))
    assert cos > 0.999, cos
    # the pickled model of the fused run loads through AMT's path (amt.py:24-27) and reproduces the live model
    with open(os.path.join(df.d_out, 'model_002_000.pkl'), 'rb') as f:
        m2 = pickle.load(f).to(dev).eval()
    x = O.synth_spec(2, cfg, salt=3).to(dev) * 0.5
    df.model.eval()
    with torch.no_grad():
        for a, b in zip(m2(x), df.model(x)):
            assert torch.equal(a, b)


def test_m_training_last_step_valid_with_metrics(dev, tmp_path, monkeypatch, capsys):
    """m_training.py (7-3), on by default: valid(metrics=True) prints the reference's three lines, writes test_performance.json with its
    three keys (train.py:235-251) and returns the same (loss sum, batches) as metrics=False; fast (reference criteria) and compat path.
    The scores are the reference's degenerate ones restated (every cell of a sigmoid output is an 'onset'): checked against the scorer
    run on the same posteriors by hand."""
    import json
    from hftt_hip.trainer import FusedAdam
    from training import train
    from evaluation.metrics import reshape_for_mir_eval, transcription_evaluate
    d = _Driver(dev, str(tmp_path / 'run'), MINI, lambda ps, lr: FusedAdam(ps, lr=lr), dropout=0.0)
    os.makedirs(d.d_out, exist_ok=True)
    d.epoch(0)
    monkeypatch.chdir(tmp_path)                                   # the reference writes the JSON into the working directory
    loss_test, loader = d.test()
    out = capsys.readouterr().out
    with open(tmp_path / 'test_performance.json') as f:
        perf = json.load(f)
    assert set(perf) == {'precision', 'recall', 'f1'}
    for k, label in (('precision', 'Precision:'), ('recall', 'Recall:'), ('f1', 'F1:')):
        assert '%s %s' % (label, perf[k]) in out, out
    # by hand: same posteriors, same pooling per batch, mean over batches
    d.model.eval()
    acc = np.zeros(3)
    with torch.no_grad():
        for spec, on, _off, _mpe, _vel in loader:
            o = d.model(spec.to(dev))
            ei, ep = reshape_for_mir_eval(o[5].cpu().numpy(), o[6].cpu().numpy())
            ri, rp = reshape_for_mir_eval(on.numpy(), on.numpy())
            assert len(ei) == o[5].numel()                        # a sigmoid output is never zero: one 'note' per cell
            sc = transcription_evaluate(ri, rp, ei, ep)
            acc += [sc['Precision'], sc['Recall'], sc['F-measure']]
    acc /= len(loader)
    assert perf['precision'] == pytest.approx(acc[0], abs=1e-12) and perf['recall'] == pytest.approx(acc[1], abs=1e-12)
    assert perf['f1'] == pytest.approx(acc[2], abs=1e-12)
    assert 0.0 < perf['precision'] < 0.2 and perf['recall'] > 0.5         # README.md:6-19's shape: P ~ 0.01, R ~ 0.95
    lv, n = train.valid(d.model, loader, *d.crits, 1.0, 1.0, dev, metrics=False)
    assert n == len(loader) and lv / n == pytest.approx(loss_test, rel=1e-6)
    # compatibility path (a criterion that is not the reference's) takes the same branch
    crits = list(d.crits)
    crits[0] = nn.BCELoss(reduction='mean', weight=None)
    crits[3] = nn.CrossEntropyLoss(label_smoothing=1e-9)
    os.remove(tmp_path / 'test_performance.json')
    lc, nc = train.valid(d.model, loader, *crits, 1.0, 1.0, dev, metrics=True)
    with open(tmp_path / 'test_performance.json') as f:
        assert json.load(f) == perf
    assert lc / nc == pytest.approx(loss_test, rel=1e-4)


def test_resume_is_bit_identical_with_dropout_on(dev, tmp_path):
    """(6) resume: a run stopped after epoch 0 and resumed from model_000_000.dat continues exactly like the uninterrupted run --
    parameters, both Adam moments, the step count, the scheduler and the position of the dropout stream (FusedAdam.state_dict carries it)."""
    from hftt_hip.trainer import FusedAdam
    cfg = MINI
    mk = lambda ps, lr: FusedAdam(ps, lr=lr)       # noqa: E731
    a = _Driver(dev, str(tmp_path / 'a'), cfg, mk, dropout=0.1)
    os.makedirs(a.d_out, exist_ok=True)
    a.epoch(0)
    la = a.epoch(1)
    b = _Driver(dev, str(tmp_path / 'a'), cfg, mk, dropout=0.1, seed=999)      # a fresh process: different init, then (6) resume
    ck = b.resume(0)
    assert ck['epoch'] == 0 and 'hftt_step_counter' in ck['optimizer_dict']
    lb = b.epoch(1)
    assert la == lb, (la, lb)
    assert torch.equal(a.params(), b.params())
    assert torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq) and a.optimizer.step_count == b.optimizer.step_count
    # (the scheduler is NOT compared: the reference saves the checkpoint (7-4) before scheduler.step (7-6), so a resumed run's scheduler has
    # seen one validation loss less than the uninterrupted one -- m_training.py:374-392 vs :437)
    assert a.optimizer.param_groups[0]['lr'] == b.optimizer.param_groups[0]['lr']


# ---------------------------------------------------------------------------------------------------------------------------------
# reference-made checkpoints (generated by tests/golden/make_golden_r2.py with the REFERENCE classes)
def test_reference_checkpoint_runs_on_the_gpu(dev):
    from model.amt import AMT
    g = np.load(os.path.join(G, 'ref_ckpt.npz'))
    c = {str(k): int(v) for k, v in zip(g['cfg_keys'], g['cfg'])}
    config = {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': c['n_bin'], 'n_bins': c['n_bin'], 'log_offset': 1e-8},
              'input': {'margin_b': c['n_margin'], 'margin_f': c['n_margin'], 'num_frame': c['n_frame'], 'min_value': -18.420681},
              'midi': {'note_min': 21, 'note_max': 21 + c['n_note'] - 1, 'num_note': c['n_note'], 'num_velocity': c['n_velocity']}}
    amt = AMT(config, os.path.join(G, 'ref_ckpt_model.pkl'), batch_size=2)          # pickle.load -> .to(device) -> .eval()  (amt.py:24-27)
    with torch.no_grad():
        outs = amt.model(torch.from_numpy(g['x']).to(dev))
    worst = 0.0
    for k, o in enumerate(outs):
        ref = g['out%d' % k]
        assert tuple(o.shape) == ref.shape
        worst = max(worst, float(np.abs(o.cpu().numpy() - ref).max()))
    print('reference checkpoint: model outputs max abs err %.2e' % worst)
    assert worst < 1e-3
    tr = amt.transcript(g['feature'])
    for k, o in enumerate(tr):
        ref = g['tr%d' % k]
        assert o.shape == ref.shape and o.dtype == ref.dtype
        if k % 4 == 3:
            assert (o != ref).mean() < 0.01
        else:
            assert np.abs(o - ref).max() < 1e-3


def test_reference_dat_resume_then_one_step_equals_torch_adam(dev):
    """optimizer_dict of the reference (torch.optim.Adam after one step) -> FusedAdam; the next step equals torch.optim.Adam's
    (same gradient fed to both)."""
    from hftt_hip.trainer import FusedAdam, TrainStep
    g = np.load(os.path.join(G, 'ref_ckpt.npz'))
    c = O.HfttConfig(**{str(k): int(v) for k, v in zip(g['cfg_keys'], g['cfg'])})
    ck = torch.load(os.path.join(G, 'ref_ckpt_model.dat'), map_location='cpu', weights_only=False)
    model = util.build_model(c, 1, dropout=0.0).to(dev)
    model.load_state_dict(ck['model_dict'])
    opt = FusedAdam(model.parameters(), lr=3e-4)
    opt.load_state_dict(ck['optimizer_dict'])                  # before the first forward: state waits for the engine
    step = TrainStep(model, optimizer=opt)
    model.train()
    x = O.synth_spec(2, c, salt=5) * 0.5
    labels = O.synth_labels(2, c, salt=6)
    step.forward_backward(x.to(dev), *[t.to(dev).contiguous() for t in labels])
    grads = [gv.clone().cpu() for gv in step.engine.grad_views()]
    # torch.optim.Adam on the CPU, same parameters / state / gradients
    ref_params = [nn.Parameter(ck['model_dict'][n].clone()) for n, _ in model.named_parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=3e-4)
    ref_opt.load_state_dict(ck['optimizer_dict'])
    for p, gr in zip(ref_params, grads):
        p.grad = gr
    ref_opt.step()
    opt.step()
    assert opt.step_count == 2 and opt.param_groups[0]['lr'] == 1e-4
    worst = max((p.detach().cpu() - r.detach()).abs().max().item() for p, r in zip(model.parameters(), ref_params))
    assert worst < 2e-7, worst
    sd = opt.state_dict()
    for i, r in ref_opt.state_dict()['state'].items():
        assert float(sd['state'][i]['step']) == float(r['step']) == 2.0
        assert (sd['state'][i]['exp_avg_sq'].cpu() - r['exp_avg_sq']).abs().max().item() <= 1e-6 * r['exp_avg_sq'].abs().max().item() + 1e-12


# ---------------------------------------------------------------------------------------------------------------------------------
# data parallel: two real processes, one engine each (both on cuda:0 of the one-GPU box), gloo as the transport
def _ddp_compat_worker(rank, world, port, tmp):
    """The reference's loop unchanged (torch.optim.Adam -> the autograd / compatibility path of training.train) under world 2, with the
    optimizer told to KEEP gradient tensors across steps (zero_grad(set_to_none=False)): p.grad then is a tensor of its own, not a view of
    the engine's flat gradient buffer, and the reduced buffer must still be what the optimizer steps on."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import functools
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from training import train as T
    from training.dataset import MyDataset, DeviceClipStore
    from corpus.make_dataset import synth_store
    dev = torch.device('cuda:0')
    cfg = MINI
    conf = _ds_config(cfg)
    store = synth_store(conf, [70, 45], seed=11)
    ds = MyDataset.from_arrays(store['feature'], store['label_onset'], store['label_offset'], store['label_mpe'], store['label_velocity'],
                               store['idx'], conf, 4)
    clips = DeviceClipStore(ds, dev)
    model = util.build_model(cfg, 100 + rank, dropout=0.0).to(dev)            # different init per rank: the broadcast must fix it
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.zero_grad = functools.partial(opt.zero_grad, set_to_none=False)
    for p in model.parameters():                                               # gradients that exist before the first backward
        p.grad = torch.zeros_like(p)
    crits = [nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss(), nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss()]
    loader = clips.loader(2, rank=rank, world=world)
    losses = [T.train(model, loader, opt, *crits, 1.0, 1.0, dev, False) for _ in range(2)]
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])                                        # the ranks stay in lockstep
    if rank == 0:
        torch.save({'flat': flat, 'losses': losses}, os.path.join(tmp, 'ddp_compat.pt'))
    dist.destroy_process_group()


def test_two_process_compat_path_stays_in_lockstep(dev, tmp_path):
    """... and equals the fast path's result on the same shards (FusedAdam + fused loss): same losses, same parameters to Adam's noise."""
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_ddp_compat_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_ddp_worker, args=(2, port + 1, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / 'ddp_compat.pt', weights_only=False)
    b = torch.load(tmp_path / 'ddp.pt', weights_only=False)
    for x, y in zip(a['losses'], b['losses']):
        assert abs(x - y) < 1e-4 * abs(y), (a['losses'], b['losses'])
    assert (a['flat'] - b['flat']).abs().mean().item() < 2e-5


def _ddp_worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hftt_hip.trainer import FusedAdam
    from hftt_hip import ddp
    from training import train as T
    from training.dataset import MyDataset, DeviceClipStore
    from corpus.make_dataset import synth_store
    dev = torch.device('cuda:0')
    cfg = MINI
    conf = _ds_config(cfg)
    store = synth_store(conf, [70, 45], seed=11)
    ds = MyDataset.from_arrays(store['feature'], store['label_onset'], store['label_offset'], store['label_mpe'], store['label_velocity'],
                               store['idx'], conf, 4)
    clips = DeviceClipStore(ds, dev)
    model = util.build_model(cfg, 100 + rank, dropout=0.0).to(dev)            # different init per rank: the broadcast must fix it
    opt = FusedAdam(model.parameters(), lr=1e-3)
    crits = [nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss(), nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss()]
    loader = clips.loader(2, rank=rank, world=world)
    losses = [T.train(model, loader, opt, *crits, 1.0, 1.0, dev, False) for _ in range(2)]
    lv, n = T.valid(model, loader, *crits, 1.0, 1.0, dev, metrics=False)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])                                        # the ranks stay in lockstep
    assert n == world * len(loader)
    if rank == 0:
        torch.save({'flat': flat, 'losses': losses, 'valid': (lv, n), 'ids': [c.tolist() for c in loader.chunks]}, os.path.join(tmp, 'ddp.pt'))
    dist.destroy_process_group()


def test_two_process_data_parallel_step_equals_the_global_batch(dev, tmp_path):
    """training.train.train under world 2 (TrainStep with a live FlatGradSync: overlapped bucket all-reduce + 1/world folded into Adam)
    == ONE process stepping on the concatenated batches (mean-reduction losses)."""
    from hftt_hip.trainer import FusedAdam, TrainStep
    from training.dataset import MyDataset, DeviceClipStore
    from corpus.make_dataset import synth_store
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = torch.load(tmp_path / 'ddp.pt', weights_only=False)
    cfg = MINI
    conf = _ds_config(cfg)
    store = synth_store(conf, [70, 45], seed=11)
    ds = MyDataset.from_arrays(store['feature'], store['label_onset'], store['label_offset'], store['label_mpe'], store['label_velocity'],
                               store['idx'], conf, 4)
    clips = DeviceClipStore(ds, dev)
    model = util.build_model(cfg, 100, dropout=0.0).to(dev)                     # rank 0's init
    step = TrainStep(model, optimizer=FusedAdam(model.parameters(), lr=1e-3))
    model.train()
    n_steps = len(r['ids'])
    assert n_steps >= 3
    losses = []
    for _ in range(2):
        tot = 0.0
        for s in range(n_steps):
            ids0 = r['ids'][s]
            ids = sorted(ids0 + [i + 1 for i in ids0])                            # rank 0 holds clips 0,2,4..; rank 1 the odd ones
            b = clips.batch(ids)
            tot += float(step(b[0], b[1].contiguous(), b[2].contiguous(), b[3].contiguous(), b[4].contiguous())[0])
        losses.append(tot / n_steps)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    for a, b in zip(r['losses'], losses):
        assert abs(a - b) < 1e-4 * abs(b), (r['losses'], losses)
    assert (r['flat'] - flat).abs().mean().item() < 2e-5
    assert (r['flat'] - flat).abs().max().item() <= 2.1e-3 * 2 * n_steps            # Adam: ~lr per step where a noise-level gradient flips sign


# ---------------------------------------------------------------------------------------------------------------------------------
# inference scatter with real engines: two processes, one engine each on the one GPU, gloo as the transport
def _amt_worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from model.amt import AMT
    g = np.load(os.path.join(G, 'ref_ckpt.npz'))
    c = {str(k): int(v) for k, v in zip(g['cfg_keys'], g['cfg'])}
    config = {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': c['n_bin'], 'n_bins': c['n_bin'], 'log_offset': 1e-8},
              'input': {'margin_b': c['n_margin'], 'margin_f': c['n_margin'], 'num_frame': c['n_frame'], 'min_value': -18.420681},
              'midi': {'note_min': 21, 'note_max': 21 + c['n_note'] - 1, 'num_note': c['n_note'], 'num_velocity': c['n_velocity']}}
    amt = AMT(config, os.path.join(G, 'ref_ckpt_model.pkl'), batch_size=1, device='cuda:0')          # rank / world from torch.distributed
    assert (amt.rank, amt.world) == (rank, world)
    outs = amt.transcript(g['feature'])                        # 4 clips of batch 1: this rank runs two of them; rank 0 collects by clip index on the host
    outs_s = amt.transcript_stride(g['feature'], 3)
    assert amt.gather == 'host' 
    np.savez(os.path.join(tmp, 'amt_rank%d.npz' % rank), **{'o%d' % k: o for k, o in enumerate(outs)}, **{'s%d' % k: o for k, o in enumerate(outs_s)})
    dist.destroy_process_group()


def test_two_process_inference_scatter_equals_one_engine(dev, tmp_path):
    """model/amt.py under world 2 (clip batches dealt round-robin, replicas only; rank 0 gathers the shards by clip index on the host -- no
    device collective) is BIT-EQUAL, on rank 0, to the single-engine run of the same reference-made checkpoint (amt.py:86-113 stitching
    order); rank 1 holds exactly its own clips of it."""
    from model.amt import AMT
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_amt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(os.path.join(G, 'ref_ckpt.npz'))
    c = {str(k): int(v) for k, v in zip(g['cfg_keys'], g['cfg'])}
    config = {'feature': {'sr': 16000, 'hop_sample': 256, 'mel_bins': c['n_bin'], 'n_bins': c['n_bin'], 'log_offset': 1e-8},
              'input': {'margin_b': c['n_margin'], 'margin_f': c['n_margin'], 'num_frame': c['n_frame'], 'min_value': -18.420681},
              'midi': {'note_min': 21, 'note_max': 21 + c['n_note'] - 1, 'num_note': c['n_note'], 'num_velocity': c['n_velocity']}}
    amt = AMT(config, os.path.join(G, 'ref_ckpt_model.pkl'), batch_size=1, rank=0, world=1)
    one = amt.transcript(g['feature'])
    one_s = amt.transcript_stride(g['feature'], 3)
    T = c['n_frame']
    for rank in range(2):
        r = np.load(tmp_path / ('amt_rank%d.npz' % rank))
        for k in range(8):
            assert r['o%d' % k].dtype == one[k].dtype and r['o%d' % k].shape == one[k].shape
            if rank == 0:
                assert np.array_equal(r['o%d' % k], one[k]), (rank, k)
                assert np.array_equal(r['s%d' % k], one_s[k]), (rank, 's', k)
            else:                                              # clips 1 and 3 (batch 1, round-robin) and nothing else
                mine = (np.arange(one[k].shape[0]) // T) % 2 == 1
                assert np.array_equal(r['o%d' % k][mine], one[k][mine]) and not r['o%d' % k][~mine].any(), (rank, k)


@pytest.mark.parametrize('config', ['tiny', 'paper'])
def test_one_rank_rccl_rehearsal_of_the_bench_ddp_branch(dev, config):
    """bench.py's N > 1 branch -- init_process_group('nccl') = RCCL, parameter broadcast, the three bucket all-reduces on the side stream,
    the barriers, the all-gathered device list in the JSON line -- with a ONE-rank group on the one GPU of this box (the multi-GPU node
    only exists at the driver's round end).  'paper' is BASELINE config 4's own size: three buckets of the 22 MB flat
    gradient against the persistent one-workgroup-per-CU kernels of the paper-size backward."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HFTT_BENCH_FORCE_DDP='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(39500 + (os.getpid() % 2000) + (7 if config == 'paper' else 0)))
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, 'bench.py'), '--config', config, '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--no-extras', '--no-profile'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['value'] > 0 and line['n_gpus'] == 1
    col = line['collective']
    assert col['backend'] == 'nccl' and col['ranks'] == 1 and len(col['devices']) == 1
    # the three gradient buckets in release order (time decoder + heads B, frequency decoder + heads A, encoder): the all-reduces of the
    # first two were enqueued on the side stream while the backward still had the encoder to go, and finished before its last kernel
    assert len(col['buckets']) == 3 and col['buckets'][0]['range'][0] > col['buckets'][1]['range'][0] > col['buckets'][2]['range'][0] == 0
    print('one-rank RCCL rehearsal, %s: %.1f clips/s, %.2f ms/step, buckets %s' % (config, line['value'], line['ms_per_step'], json.dumps(col['buckets'])))
    out = os.path.join(util.ROOT, 'gpurun_out')
    if os.path.isdir(out):                              # (the record DESIGN.md section 7 quotes)
        json.dump({'config': config, 'clips_per_s': line['value'], 'ms_per_step': line['ms_per_step'], 'collective': col},
                  open(os.path.join(out, 'r06_rccl_rehearsal_%s.json' % config), 'w'), indent=1)
    assert col['buckets'][0]['ms_before_backward_end'] > 0.0 and col['buckets'][1]['ms_before_backward_end'] > 0.0, col['buckets']
    assert col['early_buckets_hidden'] is True
    if config == 'paper':
        flat = col['buckets'][0]['range'][1]
        assert flat >= 5516574 and line['config']['global_batch'] == 8       # the 22 MB flat gradient of the paper-size model (165 tensors + alignment)


@pytest.mark.parametrize('config', ['tiny', 'paper'])
def test_two_rank_bench_on_one_gpu(dev, config):
    """bench.py under torch.distributed.run with TWO ranks sharing this box's GPU (HFTT_BENCH_SHARE_GPU=1: gloo collectives, both ranks on
    cuda:0): the launch contract of the driver's N > 1 runs -- RANK / LOCAL_RANK / WORLD_SIZE from the environment, the barrier + max-over-
    ranks timing, one JSON line from rank 0 with the all-gathered device list -- fails here for a reason instead of on the 8-GPU node.
    'paper': BASELINE config 4 at its own size, 2 x batch 8 (two ~17 GB workspaces on the one 288 GB device)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HFTT_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = str(41000 + (os.getpid() % 2000) + (7 if config == 'paper' else 0))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', port,
                        os.path.join(util.ROOT, 'bench.py'), '--gpus', '2', '--config', config, '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--no-extras', '--no-profile'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines                      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['global_batch'] == 16 and line['config']['parallelism'] == 'dp2'
    assert line['collective']['ranks'] == 2 and sorted(d['rank'] for d in line['collective']['devices']) == [0, 1]
    assert len(line['collective']['buckets']) == 3 and ('%s-size' % config) in line['config']['workload']
    print('two ranks on one GPU, %s: %.1f clips/s (both ranks share the device; gloo moves the buckets through the host)' % (config, line['value']))
