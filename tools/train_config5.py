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
import argparse, json, os, pickle, sys, tempfile, time
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
        t0 = time.time()
        tmp = tempfile.mkdtemp()
        model = bench.build_model(cfg, args.seed, 0.1, 'cpu')
        if args.init:
            with open(args.init, 'rb') as fh:
                model.load_state_dict(pickle.load(fh).state_dict())
            log['init'] = args.init
        with open(os.path.join(tmp, 'init.pkl'), 'wb') as fh:
            pickle.dump(model, fh, protocol=4)
        fe = AMT(config, os.path.join(tmp, 'init.pkl'), batch_size=1)              # (front end only)
        feats, labs = [], []
        for i in range(args.files):
            notes = SA.pluck_notes(2000 + i)
            f = fe.wave2feature(SA.pluck_wave(notes, device=dev).unsqueeze(0), SA.SR).numpy()
            lab = note2label_arrays(config, notes)
            n = f.shape[0]
            lab = {k: (np.concatenate([v, np.zeros((n - len(v),) + v.shape[1:], v.dtype)]) if len(v) < n else v[:n]) for k, v in lab.items()}
            feats.append(f); labs.append(lab)
        store = assemble_store(feats, labs, config)
        ds = MyDataset.from_arrays(store['feature'], store['label_onset'], store['label_offset'], store['label_mpe'], store['label_velocity'],
                                   store['idx'], config, 8)
        clips = DeviceClipStore(ds, dev)
        log['corpus'] = {'files': args.files, 'frames': int(store['feature'].shape[0]), 'clips': len(clips), 'seconds_to_build': round(time.time() - t0, 1)}
        # ---- training ----
        from hftt_hip.trainer import TrainStep
        model = model.to(dev)
        model.hftt_precision = args.precision
        model.train()
        ts = TrainStep(model, lr=args.lr)
        t0, step, epoch, curve = time.time(), 0, 0, []
        budget = args.minutes * 60.0
        acc = torch.zeros(9, device=dev)
        done = False
        while not done:
            for b in clips.loader(args.batch, shuffle=True, seed=args.seed + epoch, drop_last=True):
                acc += ts(b[0], *b[1:])
                step += 1
                if step % 250 == 0:
                    l = (acc / 250).tolist(); acc.zero_()
                    curve.append((step, round(l[0], 4)))
                    print('step %6d  loss %.4f  (%.0f s, %.0f clips/s)' % (step, l[0], time.time() - t0, step * args.batch / (time.time() - t0)), flush=True)
                    if l[0] != l[0]:
                        log['diverged_at_step'] = step
                    if l[0] != l[0] or (args.steps and step >= args.steps) or (not args.steps and time.time() - t0 > budget):
                        done = True
                        break
            epoch += 1
        torch.cuda.synchronize()
        log['training'] = {'steps': step, 'epochs_started': epoch, 'seconds': round(time.time() - t0, 1), 'lr': args.lr, 'batch': args.batch, 'dropout': 0.1,
                           'clips_per_s': round(step * args.batch / (time.time() - t0), 1), 'loss_curve': curve[:: max(1, len(curve) // 24)]}
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
