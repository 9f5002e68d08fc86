#!/usr/bin/env python3
"""Same-checkpoint A/B of the precision modes (GPU box): from ONE saved training state (parameters, both Adam moments, step and dropout
counters: tools/train_config5.py --save-state) continue `--steps` steps in one mode on the same batches with the same device dropout masks,
and record the loss curve, the gradient of the first batch and the parameters at the end.  Run once per mode (the modes differ in
environment switches), then `--compare` over the result files:

  python tools/ab_modes.py --state /tmp/st_20000.pt --mode x3     --out /tmp/ab_x3.pt
  python tools/ab_modes.py --state /tmp/st_20000.pt --mode parity --out /tmp/ab_parity.pt
  python tools/ab_modes.py --state /tmp/st_20000.pt --mode bf16   --out /tmp/ab_bf16.pt
  HFTT_X3_FP32_HIDDEN=1 python tools/ab_modes.py --state /tmp/st_20000.pt --mode x3 --out /tmp/ab_x3fp32h.pt
  python tools/ab_modes.py --compare parity=/tmp/ab_parity.pt x3=/tmp/ab_x3.pt bf16=/tmp/ab_bf16.pt x3_fp32_hidden=/tmp/ab_x3fp32h.pt

What the comparison answers: does the default mode's arithmetic (fp16 / bf16 operand pairs, bf16 saved copies) move a
paper-size run away from the exact-fp32 mode's -- loss trajectory, per-tensor gradient cosine, distance between the end points relative to
the distance travelled."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch


def run(args):
    import bench
    import train_config5 as T
    from corpus import synth_audio as SA
    from hftt_hip.trainer import TrainStep
    dev = torch.device('cuda:0')
    clips, _ = T.build_corpus(SA.default_config(), args.files, dev)
    model = bench.build_model(bench.CONFIGS[args.config], 77, 0.1, 'cpu').to(dev)
    model.hftt_precision = args.mode
    model.train()
    ts = TrainStep(model, lr=1e-4)
    batches = list(clips.loader(8, shuffle=True, seed=4242, drop_last=True))
    ts.forward_backward(batches[0][0], *batches[0][1:])              # binds the engine, allocates the moments
    st = T.load_state(ts, args.state)
    if args.lr > 0:
        ts.opt.param_groups[0]['lr'] = args.lr
    start = ts.engine.flat_params.clone()
    # the first batch's gradient at the checkpoint's parameters (the dropout counter is put back afterwards: the run below sees the same masks)
    ctr = int(ts.engine.step_counter)
    ts.forward_backward(batches[0][0], *batches[0][1:])
    grad0 = ts.engine.flat_grads.clone().cpu()
    ts.engine.step_counter = ctr
    names = [(n, o, k) for (n, _p, o, k) in ts.engine._bound]
    curve, acc, t0 = [], torch.zeros(9, device=dev), time.time()
    for s in range(args.steps):
        b = batches[s % len(batches)]
        acc += ts(b[0], *b[1:])
        if (s + 1) % args.every == 0:
            curve.append((s + 1, float(acc[0]) / args.every)); acc.zero_()
            print('%s step %5d loss %.5f (%.0f s)' % (args.mode, s + 1, curve[-1][1], time.time() - t0), flush=True)
    torch.save({'mode': args.mode, 'env': {k: v for k, v in os.environ.items() if k.startswith('HFTT_')}, 'state_step': st['step'], 'lr': ts.opt.param_groups[0]['lr'],
                'curve': curve, 'grad0': grad0, 'start': start.cpu(), 'end': ts.engine.flat_params.cpu(), 'names': names,
                'clips_per_s': args.steps * 8 / (time.time() - t0)}, args.out)


def compare(pairs):
    res = {k: torch.load(v, weights_only=False) for k, v in (p.split('=') for p in pairs)}
    base_k = 'parity' if 'parity' in res else next(iter(res))
    base = res[base_k]
    out = {'base': base_k, 'state_step': base['state_step'], 'lr': base['lr'], 'modes': {}}
    travelled = float((base['end'] - base['start']).double().norm())
    for k, r in res.items():
        m = {'clips_per_s': round(r['clips_per_s'], 1), 'env': r['env'], 'loss_curve': [(s, round(l, 5)) for s, l in r['curve']]}
        if k != base_k:
            m['max_rel_loss_difference'] = round(max(abs(a[1] - b[1]) / b[1] for a, b in zip(r['curve'], base['curve'])), 5)
            cos = []
            for n, o, cnt in r['names']:
                a, b = r['grad0'][o:o + cnt].double(), base['grad0'][o:o + cnt].double()
                if float(b.norm()) > 0 and float(a.norm()) > 0:
                    cos.append((float((a * b).sum() / (a.norm() * b.norm())), n))
            cos.sort()
            ga, gb = r['grad0'].double(), base['grad0'].double()
            m['first_batch_gradient'] = {'whole_vector_cosine': round(float((ga * gb).sum() / (ga.norm() * gb.norm())), 6),
                                         'per_tensor_cosine_min': [round(cos[0][0], 4), cos[0][1]], 'per_tensor_cosine_median': round(cos[len(cos) // 2][0], 6),
                                         'tensors_below_0.99': sum(1 for c, _ in cos if c < 0.99), 'tensors': len(cos),
                                         'rel_norm_difference': round(float((ga - gb).norm() / gb.norm()), 5)}
            m['end_point_distance_over_distance_travelled'] = round(float((r['end'] - base['end']).double().norm()) / travelled, 5)
        out['modes'][k] = m
    out['distance_travelled_by_base'] = round(travelled, 4)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--state', default='')
    ap.add_argument('--mode', default='x3')
    ap.add_argument('--config', default='paper')
    ap.add_argument('--steps', type=int, default=3000)
    ap.add_argument('--every', type=int, default=250)
    ap.add_argument('--files', type=int, default=48)
    ap.add_argument('--lr', type=float, default=0.0, help='0: the learning rate stored with the state')
    ap.add_argument('--out', default='/tmp/ab.pt')
    ap.add_argument('--compare', nargs='+', default=None)
    args = ap.parse_args()
    if args.compare:
        compare(args.compare)
    else:
        run(args)


if __name__ == '__main__':
    main()
