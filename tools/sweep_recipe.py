#!/usr/bin/env python3
"""Learning-rate recipes side by side on the synthetic corpus (GPU box): same corpus, same initial parameters, same batches, same dropout
masks; only the schedule differs.  Answers "which schedule leaves the loss plateau" cheaply on the small model before the paper-size run.

  python tools/sweep_recipe.py --config tiny --steps 15000 --recipes const:3e-4:0:1:0 warm:1e-3:1000:1:0 warmcos:1e-3:1000:0.1:0

A recipe is name:peak_lr:warmup_steps:final_frac:clip[:pos_scale] (tools/train_config5.py::lr_at, scale_position_embeddings_).  One JSON line with every loss curve."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench
import train_config5 as T
from corpus import synth_audio as SA


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='tiny')
    ap.add_argument('--precision', default='x3')
    ap.add_argument('--steps', type=int, default=15000)
    ap.add_argument('--files', type=int, default=48)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--seed', type=int, default=77)
    ap.add_argument('--log-every', type=int, default=500)
    ap.add_argument('--recipes', nargs='+', required=True)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    clips, corpus = T.build_corpus(SA.default_config(), args.files, dev)
    out = {'config': args.config, 'precision': args.precision, 'steps': args.steps, 'batch': args.batch, 'seed': args.seed, 'corpus': corpus, 'recipes': {}}
    for r in args.recipes:
        f = r.split(':')
        name, lr, warm, ff, clip = f[:5]
        pos = float(f[5]) if len(f) > 5 else 1.0
        model = bench.build_model(bench.CONFIGS[args.config], args.seed, 0.1, 'cpu')
        T.scale_position_embeddings_(model, pos)
        model = model.to(dev)
        ts, step, epoch, curve, secs = T.train_loop(model, clips, dev, args.precision, float(lr), steps=args.steps, batch=args.batch, seed=args.seed,
                                                   warmup=int(warm), final_frac=float(ff), clip=float(clip), log_every=args.log_every, tag=name + ' ')
        out['recipes'][name] = {'lr': float(lr), 'warmup': int(warm), 'final_frac': float(ff), 'clip': float(clip), 'pos_scale': pos, 'steps': step,
                                'clips_per_s': round(step * args.batch / secs, 1), 'loss_curve': curve}
        del ts, model
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
