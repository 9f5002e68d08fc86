#!/usr/bin/env python3
"""Headline benchmark: training clips/sec of the paper-size hFT-Transformer (d=256, ff=512, 3+3 layers, 4 heads),
batch 8 per GPU, 128-frame x 256-bin clips, dropout 0.1, on N MI355X GPUs (BASELINE.json configs[2]/[3]).

A "step" = forward + fused loss + backward + (gradient all-reduce when N > 1) + fused Adam over one synthetic batch that
is already resident in HBM.  One JSON line on rank 0 (see the driver contract in the task statement), with two extra
objects: "roofline" (dominant kernel, measured with HIP events inside the timed region) and "cpu_baseline" (the CPU
oracle = a port of the reference algorithm, timed on the host cores, rank 0 at N=1 only).

Launch: python bench.py --gpus 1 --steps 20 --warmup 5
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'nylon-amt_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import collections   # noqa: E402

import torch   # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
FWD_GFLOP_PER_CLIP = 249.44    # SURVEY.md section 8(d), paper size, forward; training = 3x


# Workload definitions of the MEASURED leg.  Nothing on this leg touches oracle/ (test infrastructure): the configurations restate
# the reference's defaults (m_training.py:55-60 tiny, hFT paper size), the model is built from the product package exactly as
# m_training.py:117-141 does, and the clips are random tensors of the MAESTRO clip contract (training/dataset.py:49-71).
# tests/test_boundary.py checks these dictionaries against the oracle's configurations so they cannot drift apart.
BenchCfg = collections.namedtuple('BenchCfg', 'n_margin n_frame n_bin cnn_channel cnn_kernel hid_dim pf_dim enc_layer dec_layer enc_head dec_head n_note n_velocity')
CONFIGS = {'paper': BenchCfg(32, 128, 256, 4, 5, 256, 512, 3, 3, 4, 4, 88, 128),
           'tiny': BenchCfg(32, 128, 256, 4, 5, 64, 128, 2, 2, 2, 2, 88, 128)}


def build_model(cfg, seed, dropout, dev):
    """m_training.py:109-141: seed, positional construction, xavier_uniform on every weight with dim > 1, .to(device)."""
    from model.model_spec2midi import Encoder_SPEC2MIDI, Decoder_SPEC2MIDI, Model_SPEC2MIDI
    torch.manual_seed(seed)
    enc = Encoder_SPEC2MIDI(cfg.n_margin, cfg.n_frame, cfg.n_bin, cfg.cnn_channel, cfg.cnn_kernel, cfg.hid_dim, cfg.enc_layer, cfg.enc_head,
                            cfg.pf_dim, dropout, 'cpu')
    dec = Decoder_SPEC2MIDI(cfg.n_frame, cfg.n_bin, cfg.n_note, cfg.n_velocity, cfg.hid_dim, cfg.dec_layer, cfg.dec_head, cfg.pf_dim, dropout, 'cpu')
    model = Model_SPEC2MIDI(enc, dec)
    for m in model.modules():
        if hasattr(m, 'weight') and m.weight is not None and m.weight.dim() > 1:
            torch.nn.init.xavier_uniform_(m.weight.data)
    return model.to(dev)


def synthetic_batch(cfg, B, seed, dev):
    """One batch of the clip contract: log-mel-like spectrogram [B, n_bin, margin+frames+margin] (N(-7, 3^2) clipped to the
    reference's [-18.42, 6] range), onset/offset targets in [0, 1], binary mpe, velocity classes (int64)."""
    g = torch.Generator().manual_seed(seed)
    W = cfg.n_frame + 2 * cfg.n_margin
    spec = (torch.randn(B, cfg.n_bin, W, generator=g) * 3.0 - 7.0).clamp_(-18.420681, 6.0)
    shp = (B, cfg.n_frame, cfg.n_note)
    active = torch.rand(shp, generator=g) < 0.05
    onset = torch.rand(shp, generator=g) * (torch.rand(shp, generator=g) < 0.02)
    offset = torch.rand(shp, generator=g) * (torch.rand(shp, generator=g) < 0.02)
    velocity = torch.randint(1, cfg.n_velocity, shp, generator=g) * active
    return spec.to(dev), (onset.to(dev).contiguous(), offset.to(dev).contiguous(), active.float().to(dev).contiguous(),
                          velocity.to(torch.int64).to(dev).contiguous())


def cpu_baseline(cfg, threads):
    """CPU oracle (port of the reference algorithm), one training step of batch 1 at the SAME model config."""
    from oracle import hftt_oracle as O       # the ONLY use of oracle/ in this file: the reported CPU baseline
    torch.set_num_threads(threads)
    ocfg = O.HfttConfig(**cfg._asdict())
    model = build_model(cfg, 1234, 0.1, 'cpu')
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    cfg = ocfg
    names = list(sd.keys())
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    x = O.synth_spec(1, cfg, salt=1)
    labels = O.synth_labels(1, cfg, salt=2)
    times = []
    for step in (1, 2):
        t0 = time.time()
        for t in sd.values():
            t.grad = None
        out = O.model_forward(sd, x, cfg, p=0.1, training=True)
        loss = O.spec2midi_loss(out, *labels)
        loss.backward()
        with torch.no_grad():
            O.adam_step([sd[k] for k in names], [sd[k].grad for k in names], m, v, step)
        times.append(time.time() - t0)
    return {'value': 1.0 / times[-1], 'unit': 'clips/s', 'cores': threads, 'kind': 'port',
            'sample': 'paper-size hFT, batch 1, fp32, dropout 0.1: 1 warm-up + 1 timed step of forward+loss+backward+Adam '
                      '(%.1f s) with the pure-PyTorch CPU oracle' % times[-1]}


def pmc_traffic_bytes(kernel_key):
    """HBM bytes per launch of `kernel_key` from the newest committed PMC summary (profiles/*_bench_pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same command, FETCH_SIZE doubled for gfx950), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_bench_pmc_traffic.json')))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))['kernels'].get(kernel_key)
        return None if k is None else (k['fetch_MB_per_launch_x2_corrected'] + k['write_MB_per_launch']) * 1e6
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='clips per GPU')
    ap.add_argument('--config', default='paper', choices=['paper', 'tiny'])
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'parity'])
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='skip per-launch HIP events (roofline object becomes null)')
    args = ap.parse_args()

    from hftt_hip.trainer import TrainStep
    from hftt_hip.profiler import LaunchProfiler

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch N>1 through torch.distributed.run' % (args.gpus, world))
    # HFTT_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box): every rank uses cuda:0 and the collective runs over gloo
    share = os.environ.get('HFTT_BENCH_SHARE_GPU') == '1'
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    cfg = CONFIGS[args.config]
    B = args.batch
    model = build_model(cfg, 1234, args.dropout, dev)
    model.hftt_precision = args.precision
    model.hftt_seed = 1234 + rank
    model.train()
    grad_sync = None
    if world > 1:
        from hftt_hip.ddp import FlatGradSync
        eng = model.hftt_engine()
        dist.broadcast(eng.flat_params, 0)
        grad_sync = FlatGradSync(eng, world)
    ts = TrainStep(model, lr=1e-4, grad_sync=grad_sync)

    # synthetic MAESTRO-format clips, resident in HBM before the timed region (per-rank shard: seed 1234 + rank)
    n_batches = 4
    data = []
    for i in range(n_batches):
        data.append(synthetic_batch(cfg, B, 1234 + 1000 * rank + i, dev))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        x, lab = data[i % n_batches]
        ts(x, *lab)
    prof = None if args.no_profile else LaunchProfiler()
    ts.engine.profiler = prof
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        x, lab = data[i % n_batches]
        loss = ts(x, *lab)
    sync()
    dt = time.perf_counter() - t0
    ts.engine.profiler = None
    loss_val = float(loss[0].item())
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    result = None
    if rank == 0:
        clips = B * world * args.steps
        value = clips / dt
        roof = None
        if prof is not None:
            summ = prof.summary()
            total_ms = sum(v['ms'] for v in summ.values())
            # dominant kernel among the calls that launch exactly one kernel (hftt_gemm_tn launches its split kernel plus a
            # slab reduce, so its event interval is not one kernel's duration and would not match rocprofv3's per-kernel average)
            single = {k: v for k, v in summ.items() if not k.startswith('gemm_tn') and v['flops'] > 0}
            key, dom = max(single.items(), key=lambda kv: kv[1]['ms'])
            avg_ms = dom['ms'] / dom['launches']
            ai = dom['flops'] / max(dom['bytes'], 1.0)
            peak_tf = PEAK_BF16_TFLOPS if args.precision == 'bf16' else PEAK_F32_TFLOPS
            ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
            if dom['flops'] > 0 and ai >= ridge:
                ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
                roof = {'bound': 'mfma', 'achieved': ach, 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': ach / peak_tf, 'traffic': None}
            else:
                ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9 if dom['bytes'] > 0 else 0.0
                roof = {'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBS, 'traffic': None}
            roof['traffic'] = pmc_traffic_bytes(key)
            roof.update({'kernel': key, 'launches_per_step': dom['launches'] / args.steps, 'avg_launch_ms': avg_ms,
                         'share_of_step_device_time': dom['ms'] / max(total_ms, 1e-9),
                         'algorithmic_flops_per_launch': dom['flops'] / dom['launches'], 'algorithmic_bytes_per_launch': dom['bytes'] / dom['launches'],
                         'tflops': dom['flops'] / (dom['ms'] * 1e-3) / 1e12})
            top = sorted(summ.items(), key=lambda kv: -kv[1]['ms'])[:12]
            kernels = [{'kernel': k, 'ms_per_step': v['ms'] / args.steps, 'launches_per_step': v['launches'] / args.steps,
                        'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else None,
                        'gbs': (v['bytes'] / (v['ms'] * 1e-3) / 1e9) if v['bytes'] else None} for k, v in top]
        else:
            kernels = None
        result = {
            'metric': 'training clips/sec (128-frame x 256-bin)', 'value': value, 'unit': 'clips/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.precision == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': '%s-size hFT-Transformer training step (d=%d, ff=%d, %d+%d layers, %d heads), batch %d clips/GPU, '
                                   'dropout %.2f, forward+loss+backward+Adam' % (args.config, cfg.hid_dim, cfg.pf_dim, cfg.enc_layer, cfg.dec_layer,
                                                                                cfg.enc_head, B, args.dropout),
                       'global_batch': B * world, 'frames': cfg.n_frame, 'bins': cfg.n_bin, 'parallelism': 'dp%d' % world,
                       'precision_mode': args.precision},
            'model_tflops': 3 * FWD_GFLOP_PER_CLIP * value / 1e3 if args.config == 'paper' else None,
            'final_loss': loss_val,
            'roofline': roof,
            'kernels': kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            result['cpu_baseline'] = cpu_baseline(cfg, threads)
        else:
            result['cpu_baseline'] = None
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
