#!/usr/bin/env python3
"""Train the path on the synthetic plucked-string corpus and score config 5 with weights that mean something (GPU box).

  python tools/train_config5.py --config tiny --minutes 6 --out gpurun_out/config5_tiny.pkl

Corpus: `--files` one-minute files from corpus.synth_audio (seeds 2000.., never the scored file's seed 1234) -> HIP log-mel
(model.amt.AMT.wave2feature) -> frame labels (corpus.conv_note2label.note2label_arrays, the reference's label recipe) -> one MAESTRO-format
store (corpus.make_dataset.assemble_store) resident in HBM (training.dataset.DeviceClipStore).  Training: the product's own step
(hftt_hip.trainer.TrainStep: forward + fused loss + backward + fused Adam) in the chosen precision mode, dropout 0.1, batch 8, until the
time budget is spent.  The model is pickled the way m_training.py:372-373 does it (whole module, protocol 4).

Scoring (config 5): the seed-1234 minute -> log-mel -> 30 clips through AMT.transcript -> mpe2note -> note-F1 (onset, 50 ms) and frame-F1
against the GENERATING notes, in x3, bf16 and parity mode, plus the frame-level agreement of the modes with each other.  One JSON line."""
import argparse, json, math, os, pickle, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import numpy as np
import torch
import bench
from corpus import synth_audio as SA
from corpus.conv_note2label import note2label_arrays
from corpus.make_dataset import assemble_store
from training.dataset import MyDataset, DeviceClipStore
from model.amt import AMT
from evaluation.metrics import note_metrics, frame_metrics


def build_corpus(config, files, dev, n_slice=8):
    """`files` one-minute plucked-string files (seeds 2000..) -> HIP log-mel -> reference-recipe labels -> one MAESTRO-format store in HBM"""
    t0 = time.time()
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, 'init.pkl'), 'wb') as fh:
        pickle.dump(bench.build_model(bench.CONFIGS['tiny'], 1, 0.1, 'cpu'), fh, protocol=4)
    fe = AMT(config, os.path.join(tmp, 'init.pkl'), batch_size=1)              # (front end only)
    feats, labs = [], []
    for i in range(files):
        notes = SA.pluck_notes(2000 + i)
        f = fe.wave2feature(SA.pluck_wave(notes, device=dev).unsqueeze(0), SA.SR).numpy()
        lab = note2label_arrays(config, notes)
        n = f.shape[0]
        lab = {k: (np.concatenate([v, np.zeros((n - len(v),) + v.shape[1:], v.dtype)]) if len(v) < n else v[:n]) for k, v in lab.items()}
        feats.append(f); labs.append(lab)
    store = assemble_store(feats, labs, config)
    ds = MyDataset.from_arrays(store['feature'], store['label_onset'], store['label_offset'], store['label_mpe'], store['label_velocity'],
                               store['idx'], config, n_slice)
    clips = DeviceClipStore(ds, dev)
    return clips, {'files': files, 'frames': int(store['feature'].shape[0]), 'clips': len(clips), 'seconds_to_build': round(time.time() - t0, 1)}


def scale_position_embeddings_(model, scale):
    """multiply the three position tables (encoder bins, decoder notes, decoder frames) by `scale` -- an INITIALISATION choice of this tool.
    Why: the reference applies xavier_uniform_ to its nn.Embedding tables too (m_training.py:31-33: |w| <= 0.11 at 256 x 256) and adds them to
    token embeddings multiplied by sqrt(hid_dim) (model_spec2midi.py:95,190) whose entries are in the hundreds on raw log-mel input, so at
    initialisation the position of a bin / frame is 1e-3 of what LayerNorm sees and the model sits on the predict-the-prior plateau until
    Adam has grown the tables (tens of thousands of steps).  Measured with the fp32 CPU oracle itself: same
    seed, 600 steps: held-out ranking AUC 0.61 with the reference's tables, 0.99 with the tables x 30."""
    if scale == 1.0:
        return
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'pos_embedding' in name:
                p.mul_(scale)


def lr_at(step, lr, warmup=0, total=0, final_frac=1.0):
    """learning rate of optimizer step `step` (1-based): linear warm-up from lr / warmup to lr over `warmup` steps, then constant, or -- with
    `total` and `final_frac` < 1 -- a half cosine from lr down to lr * final_frac at step `total` (held there afterwards).  This schedule is
    the TOOL's (the reference's loop has ReduceLROnPlateau only, m_training.py:147): a post-LN transformer of the paper's width does not
    leave the loss plateau at a constant rate in the time one GPU call allows."""
    if warmup and step <= warmup:
        return lr * step / warmup
    if total and final_frac < 1.0:
        x = min(1.0, (step - warmup) / max(1, total - warmup))
        return lr * (final_frac + (1.0 - final_frac) * 0.5 * (1.0 + math.cos(math.pi * x)))
    return lr


def clip_flat_gradient_(flat, max_norm):
    """global-norm clipping of the engine's flat gradient without a host sync (torch ops on the flat buffer: plumbing beside the fused Adam)"""
    n = torch.linalg.vector_norm(flat)
    flat.mul_(torch.clamp(max_norm / (n + 1e-6), max=1.0))
    return n


def train_loop(model, clips, dev, precision, lr, steps=0, minutes=0.0, batch=8, seed=77, warmup=0, final_frac=1.0, clip=0.0, log_every=250,
               on_step=None, tag=''):
    """the product's own training step until `steps` (or the time budget); returns (TrainStep, steps done, epochs started, loss curve, seconds)"""
    from hftt_hip.trainer import TrainStep
    model.hftt_precision = precision
    model.train()
    ts = TrainStep(model, lr=lr)
    group = ts.opt.param_groups[0]
    t0, step, epoch, curve = time.time(), 0, 0, []
    acc = torch.zeros(9, device=dev)
    done = False
    while not done:
        for b in clips.loader(batch, shuffle=True, seed=seed + epoch, drop_last=True):
            group['lr'] = lr_at(step + 1, lr, warmup, steps, final_frac)
            if clip > 0.0:
                acc += ts.forward_backward(b[0], *b[1:])
                clip_flat_gradient_(ts.engine.flat_grads, clip)
                with torch.cuda.device(ts.engine.device):
                    ts.opt.step()
            else:
                acc += ts(b[0], *b[1:])
            step += 1
            if on_step is not None:
                on_step(ts, step)
            if step % log_every == 0:
                l = (acc / log_every).tolist(); acc.zero_()
                curve.append((step, round(l[0], 4)))
                print('%sstep %6d  loss %.4f  lr %.2e  (%.0f s, %.0f clips/s)' % (tag, step, l[0], group['lr'], time.time() - t0, step * batch / (time.time() - t0)), flush=True)
                if l[0] != l[0] or (steps and step >= steps) or (not steps and time.time() - t0 > minutes * 60.0):
                    done = True
                    break
        epoch += 1
    torch.cuda.synchronize()
    return ts, step, epoch, curve, time.time() - t0


def save_state(ts, step, path):
    eng, opt = ts.engine, ts.opt
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({'step': step, 'flat_params': eng.flat_params.cpu(), 'exp_avg': opt.exp_avg.cpu(), 'exp_avg_sq': opt.exp_avg_sq.cpu(),
                'adam_steps': opt.step_count, 'dropout_counter': int(eng.step_counter), 'lr': opt.param_groups[0]['lr']}, path)
    print('state of step %d -> %s' % (step, path), flush=True)


def load_state(ts, path):
    st = torch.load(path, weights_only=False)
    eng, opt = ts.engine, ts.opt
    eng.flat_params.copy_(st['flat_params']); opt.exp_avg.copy_(st['exp_avg']); opt.exp_avg_sq.copy_(st['exp_avg_sq'])
    opt.step_count = int(st['adam_steps']); eng.step_counter = int(st['dropout_counter'])
    opt.param_groups[0]['lr'] = st['lr']
    eng._prepared_frozen = False
    return st


def score(pkl, precision, dev, notes, wave, config):
    """config 5 on one precision mode -> (line fragment, mpe posteriorgram)"""
    amt = AMT(config, pkl, batch_size=32)
    amt.model.hftt_precision = precision
    for _ in range(2):                                     # (first pass builds plans and workspaces)
        t0 = time.time()
        feat = amt.wave2feature(wave.unsqueeze(0), SA.SR)
        torch.cuda.synchronize(); t1 = time.time()
        outs = amt.transcript(feat.numpy())
        torch.cuda.synchronize(); t2 = time.time()
        est = amt.mpe2note(a_onset=outs[4], a_offset=outs[5], a_mpe=outs[6], a_velocity=outs[7])
        t3 = time.time()
    n_clips = -(-feat.shape[0] // 128)
    roll = SA.reference_roll(notes, feat.shape[0])
    nm = note_metrics(notes, est)
    fm = frame_metrics(roll, outs[6], threshold=0.5)
    return ({'clips': n_clips, 'clips_per_s_model': round(n_clips / (t2 - t1), 1), 'seconds': {'logmel': round(t1 - t0, 4), 'model': round(t2 - t1, 4), 'mpe2note_cpu': round(t3 - t2, 4)},
             'clips_per_s_end_to_end': round(n_clips / (t3 - t0), 1), 'n_ref_notes': len(notes), 'n_est_notes': len(est),
             'note': {k: round(float(v), 4) for k, v in nm.items() if k != 'matching'},
             'frame': {k: round(float(v), 4) for k, v in fm.items()}}, outs[6])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='tiny', choices=['tiny', 'paper'])
    ap.add_argument('--precision', default='x3')
    ap.add_argument('--minutes', type=float, default=6.0)
    ap.add_argument('--steps', type=int, default=0, help='stop after this many steps instead of the time budget')
    ap.add_argument('--files', type=int, default=48)
    ap.add_argument('--lr', type=float, default=3e-4)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--seed', type=int, default=77)
    ap.add_argument('--warmup', type=int, default=0, help='linear warm-up steps (lr_at)')
    ap.add_argument('--final-frac', type=float, default=1.0, help='< 1 with --steps: cosine decay to lr * final_frac at the last step')
    ap.add_argument('--pos-scale', type=float, default=1.0, help='multiply the position-embedding tables by this at initialisation (scale_position_embeddings_)')
    ap.add_argument('--clip', type=float, default=0.0, help='global gradient-norm clip (0 = off, as the reference)')
    ap.add_argument('--save-state', default='', help='pattern with one %%d: write the full training state there at the steps of --save-state-at')
    ap.add_argument('--save-state-at', default='')
    ap.add_argument('--out', default='gpurun_out/config5_tiny.pkl')
    ap.add_argument('--score-only', default='', help='skip training: score this pickled model')
    ap.add_argument('--init', default='', help='start from this pickled model (weights only: the optimizer state starts afresh) -- chains runs that are each bounded by the GPU call limit')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    config = SA.default_config()
    cfg = bench.CONFIGS[args.config]
    log = {'config': args.config, 'precision': args.precision}
    pkl = args.score_only
    if not pkl:
        # ---- corpus ----
        model = bench.build_model(cfg, args.seed, 0.1, 'cpu')
        scale_position_embeddings_(model, args.pos_scale)
        if args.init:
            with open(args.init, 'rb') as fh:
                model.load_state_dict(pickle.load(fh).state_dict())
            log['init'] = args.init
        clips, log['corpus'] = build_corpus(config, args.files, dev)
        # ---- training ----
        model = model.to(dev)
        saver = None
        if args.save_state:                       # full training state at chosen steps (parameters, both Adam moments, counters) for tools/ab_modes.py
            at = set(int(x) for x in args.save_state_at.split(',') if x)
            def saver(ts, step):
                if step in at:
                    save_state(ts, step, args.save_state % step)
        ts, step, epoch, curve, secs = train_loop(model, clips, dev, args.precision, args.lr, steps=args.steps, minutes=args.minutes, batch=args.batch,
                                                   seed=args.seed, warmup=args.warmup, final_frac=args.final_frac, clip=args.clip, on_step=saver)
        if curve and curve[-1][1] != curve[-1][1]:
            log['diverged_at_step'] = step
        log['training'] = {'steps': step, 'epochs_started': epoch, 'seconds': round(secs, 1), 'lr': args.lr, 'warmup': args.warmup, 'final_frac': args.final_frac,
                           'clip': args.clip, 'pos_scale': args.pos_scale, 'batch': args.batch, 'dropout': 0.1, 'seed': args.seed,
                           'clips_per_s': round(step * args.batch / secs, 1), 'loss_curve': curve[:: max(1, len(curve) // 24)]}
        model.eval()
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'wb') as fh:
            pickle.dump(model.cpu(), fh, protocol=4)              # m_training.py:372-373
        pkl = args.out
        log['checkpoint'] = {'path': args.out, 'bytes': os.path.getsize(args.out)}
    # ---- config 5 ----
    notes = SA.pluck_notes(1234)
    wave = SA.pluck_wave(notes, device=dev)
    res, mpe = {}, {}
    for mode in ('x3', 'bf16', 'parity'):
        res[mode], mpe[mode] = score(pkl, mode, dev, notes, wave, config)
    agree = {}
    for mode in ('x3', 'bf16'):
        a, b = mpe[mode] >= 0.5, mpe['parity'] >= 0.5
        agree[mode] = {'frame_decisions_differing_from_parity_mode': int((a != b).sum()), 'of': int(a.size),
                       'frame_f1_against_parity_mode': round(float(frame_metrics(b, a)['f1']) if 'f1' in frame_metrics(b, a) else -1.0, 5),
                       'max_abs_posterior_difference': round(float(np.abs(mpe[mode] - mpe['parity']).max()), 6)}
    log['config5'] = {'workload': '60 s synthetic plucked-string audio (seed 1234) -> HIP log-mel -> %d clips -> AMT.transcript -> mpe2note, scored against the generating notes' % res['x3']['clips'],
                      'modes': res, 'agreement_with_parity_mode': agree}
    print(json.dumps(log))


if __name__ == '__main__':
    main()
