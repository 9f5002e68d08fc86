#!/usr/bin/env python3
"""Headline benchmark: training clips/sec of the paper-size hFT-Transformer (d=256, ff=512, 3+3 layers, 4 heads),
batch 8 per GPU, 128-frame x 256-bin clips, dropout 0.1, on N MI355X GPUs (BASELINE.json configs[2]/[3]).

A "step" = forward + fused loss + backward + (gradient all-reduce when N > 1) + fused Adam over one synthetic batch that
is already resident in HBM.  One JSON line on rank 0 (see the driver contract in the task statement), with two extra
objects: "roofline" (dominant kernel; per-launch HIP events in a profiling pass of its own AFTER the timed region, which itself runs
without any instrumentation) and "cpu_baseline" (the CPU oracle = a port of the reference algorithm, timed on the host cores, rank 0 at
N=1 only); plus "roofline_ffn" (the fused feed-forward block hftt_ffn_res_ln_fwd on the inference plan, against the MFMA roofline),
"inference_clips_per_s", "bf16_mode" / "fp32_mfma_mode" (throughput of the other two precision modes and the output differences between
them and the benchmarked one), "compat_path_clips_per_s" (the reference's training loop unchanged: torch.optim.Adam + nn criteria +
loss.backward()) and, under N > 1, "collective" (backend, ranks and the all-gathered device list the all-reduce really saw).

Precision modes.  The benchmarked default is "x3": every product in three bf16-rate MFMA passes on split operands
(fp16 hi + lo forward, bf16 hi + lo where a gradient is an operand), fp32 tensors in HBM -- the mode whose outputs meet north_star's 1e-3
against the reference (measured 1.2e-4 on the reference-generated paper-size fixture).  "bf16" is the single-pass throughput mode (outputs
3e-2 off), "parity" the exact-fp32-MFMA mode of round 1.

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
MEASURED_BF16_MFMA_LOOP_TFLOPS = 1940.0     # (see roofline_object)
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


def host_cpu():
    """(physical cores, model string) of the host, from /proc/cpuinfo"""
    cores, model = set(), None
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name' and model is None:
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))        # the box's CPU share, when the job is pinned to fewer
    except AttributeError:
        pass
    return max(1, n), model or 'unknown'


def cpu_baseline(cfg, micro=2, accum=4, timed_steps=2, config_name='paper'):
    """The CPU oracle (a port of the reference algorithm in plain PyTorch fp32) on the host cores, SAME model config and batch as the
    measured leg: a step = `accum` micro-batches of `micro` clips with gradient accumulation (8 clips) + Adam.  One untimed micro-batch
    warms the allocator, then `timed_steps` whole steps are timed."""
    from oracle import hftt_oracle as O       # the ONLY use of oracle/ in this file: the reported CPU baseline
    physical, model_name = host_cpu()
    ocfg = O.HfttConfig(**cfg._asdict())
    model = build_model(cfg, 1234, 0.1, 'cpu')
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    names = list(sd.keys())
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    x = O.synth_spec(micro, ocfg, salt=1)
    labels = O.synth_labels(micro, ocfg, salt=2)

    def micro_batch(step):
        out = O.model_forward(sd, x, ocfg, p=0.1, training=True)
        (O.spec2midi_loss(out, *labels) / accum).backward()

    # thread count: one untimed micro-batch warms the allocator, then one micro-batch is timed at 32 / 64 / all physical cores and the
    # fastest setting runs the timed steps (the oracle's many small ops stop scaling -- and then regress -- beyond a few dozen threads)
    torch.set_num_threads(min(physical, 32))
    micro_batch(0)
    sweep = {}
    for th in sorted({min(physical, 32), min(physical, 64), physical}):
        torch.set_num_threads(th)
        for t in sd.values():
            t.grad = None
        t0 = time.time()
        micro_batch(0)
        sweep[th] = time.time() - t0
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    times = []
    for step in range(1, timed_steps + 1):
        t0 = time.time()
        for t in sd.values():
            t.grad = None
        for _ in range(accum):
            micro_batch(step)
        with torch.no_grad():
            O.adam_step([sd[k] for k in names], [sd[k].grad for k in names], m, v, step)
        times.append(time.time() - t0)
    clips = micro * accum
    return {'value': clips * len(times) / sum(times), 'unit': 'clips/s', 'cores': threads, 'kind': 'port', 'cpu_model': model_name,
            'host_physical_cores': physical, 'thread_sweep_s_per_micro_batch': {str(k): round(v, 2) for k, v in sweep.items()},
            'sample': '%s-size hFT (d=%d, ff=%d, %d+%d layers), fp32, dropout 0.1, batch %d as %d micro-batches of %d clips with gradient accumulation: 1 warm-up '
                      'micro-batch + %d timed steps of forward+loss+backward+Adam (%s s) with the pure-PyTorch CPU oracle on %d threads (the fastest of a '
                      '32 / 64 / all-physical-cores sweep over one micro-batch each)'
                      % (config_name, cfg.hid_dim, cfg.pf_dim, cfg.enc_layer, cfg.dec_layer, clips, accum, micro, len(times), ' / '.join('%.1f' % t for t in times), threads)}


_PROFILE_CONFIG = 'paper'        # set by main(): PMC summaries are quoted only from a profile of the same configuration and precision mode
_PROFILE_PRECISION = 'x3'


def _profiles_newest_first(suffix):
    """committed PMC summaries of this configuration and precision mode, newest tag first.  tools/save_profiles.py writes explicit
    `config` / `precision` fields (round 4 on); older summaries carry neither and are told apart by the kernel symbols they hold."""
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*' + suffix)), reverse=True):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if 'config' in j:
            if j['config'] == _PROFILE_CONFIG and j.get('precision', 'x3') == _PROFILE_PRECISION:
                out.append(f)
        elif (('--config tiny' in j.get('note', '')) == (_PROFILE_CONFIG == 'tiny')):
            out.append(f)          # (pre-round-4 file: pmc_traffic_bytes / pmc_busy still only quote it when it has the exact kernel symbol)
    return out


def pmc_step_bytes():
    """(HBM bytes per training step summed over every kernel, source file) from the newest committed traffic summary of this configuration
    and precision mode that carries the total, or (None, None)"""
    for f in _profiles_newest_first('_bench_pmc_traffic.json'):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get('hbm_bytes_per_step') and j.get('config') == _PROFILE_CONFIG and j.get('precision') == _PROFILE_PRECISION:
            return float(j['hbm_bytes_per_step']), os.path.relpath(f, ROOT)
    return None, None


def pmc_traffic_bytes(kernel_key, kind='bench'):
    """(HBM bytes per launch of `kernel_key`, source file) from the newest committed PMC summary of this configuration that has this
    kernel symbol (profiles/*_bench_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same command,
    FETCH_SIZE doubled for gfx950); (None, None) when no committed profile has the symbol -- another kernel's figure is never quoted."""
    for f in _profiles_newest_first('_%s_pmc_traffic.json' % kind):
        try:
            k = json.load(open(f))['kernels'].get(kernel_key)
        except Exception:
            k = None
        if k is not None:
            return (k['fetch_MB_per_launch_x2_corrected'] + k['write_MB_per_launch']) * 1e6, os.path.relpath(f, ROOT)
    return None, None


def pmc_busy(kernel_key, kind='bench'):
    """MFMA-busy fraction of `kernel_key` (as tools/save_profiles.py derives it) from the newest committed profiles/*_pmc_busy.json of this
    configuration that has the symbol, or None"""
    for f in _profiles_newest_first('_%s_pmc_busy.json' % kind):
        try:
            k = json.load(open(f))['kernels'].get(kernel_key)
        except Exception:
            k = None
        if k is not None:
            return k.get('mfma_busy')
    return None


_LIVE_PMC = None        # set by measure_pmc(): counters of THIS run (training step); _LIVE_PMC_INF: the inference plan's (fused FFN)
_LIVE_PMC_INF = None
N_SIMD = 256 * 4


def _kernel_key(n):
    """the kernel symbol as the engine's plan meta spells it (tests/test_kernel_names_gpu.py): template arguments kept, namespace and
    parameter list dropped"""
    import re
    if 'namespace)::' in n and 'at::' not in n:
        m = re.search(r'::([a-z_0-9]+(?:<[^>]*>)?)\(', n)
        return m.group(1) if m else n
    return '(other) ' + n[:60]


def dump_plans(model, B, path, plans):
    """HFTT_BENCH_PLAN_DUMP (the PMC child passes): the launch plans of this process in launch order -- [kernel symbol, shape, algorithmic bytes,
    flops] per single-kernel entry -- so that the parent can tell the launches of one symbol apart by shape (dispatch j of a symbol inside a
    step is plan entry j of that symbol)"""
    eng = model.hftt_engine()
    ws = eng._ws[B]
    out = json.load(open(path)) if os.path.exists(path) else {}
    for tag in plans:
        seq = []
        for name in (('fwd', 'bwd') if tag == 'train' else ('fwd_inf',)):
            for _, _, _, meta in ws.get(name, ws['fwd'] if name == 'fwd_inf' else []):
                if meta and 'kernel' in meta:
                    seq.append([meta['kernel'], list(meta.get('shape', ())), meta['bytes'], meta['flops']])
        out['%s:%s' % (tag, eng.precision)] = seq
    json.dump(out, open(path, 'w'))


def _pmc_pass(exe, counters, child, env, tmp, tag, timeout_s):
    """one rocprofv3 --pmc pass over `child`: the dispatches in launch order, [(kernel key, {counter: value}, duration ns)]"""
    import csv, glob, subprocess
    out = os.path.join(tmp, tag)
    r = subprocess.run([exe, '--kernel-trace', '--pmc'] + counters + ['-d', out, '-o', 'b', '--output-format', 'csv', '--'] + child,
                       cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
    files = glob.glob(os.path.join(out, '**', 'b_counter_collection.csv'), recursive=True)
    if r.returncode != 0 or not files:
        return None
    disp = {}
    for row in csv.DictReader(open(files[0])):
        d = disp.setdefault(int(row['Dispatch_Id']), [_kernel_key(row['Kernel_Name']), {}, 0.0])
        d[1][row['Counter_Name']] = d[1].get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
        if row.get('End_Timestamp') and row.get('Start_Timestamp'):
            d[2] = float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    return [disp[k] for k in sorted(disp)]


def _segments(dispatches, sep, sep_is_last):
    """cut the dispatch sequence of a child run into its steps: a step ENDS with kernel `sep` (training: adam_kernel, the last launch of a
    step) or BEGINS with it (inference: im2win_kernel, the first launch of a forward with constant weights; the last forward runs to the end of
    the process).  Whatever ran before the first step (initialisation, data synthesis, plan warm-up) belongs to no step (per-symbol
    means used to include those launches)."""
    cuts = [i for i, d in enumerate(dispatches) if d[0] == sep]
    if sep_is_last:
        return [dispatches[a + 1:b + 1] for a, b in zip(cuts[:-1], cuts[1:])]
    return [dispatches[a:b] for a, b in zip(cuts, cuts[1:] + [len(dispatches)])]


def _summarise(segs_by_pass, all_by_pass, plan):
    """segs_by_pass: {'FETCH_SIZE' | 'WRITE_SIZE' | 'busy': [step, step] (lists of dispatches)}.  -> per kernel symbol and per (symbol, shape): HBM
    bytes per launch (FETCH_SIZE doubled for gfx950 -- MI355X_MICROARCH.md -- + WRITE_SIZE; counter values are KB), MFMA-busy and clock; bytes
    per step.  Dispatch j of a symbol inside a step is plan entry j of that symbol: that is how one symbol's launches are told apart by shape."""
    import collections
    per_sym = collections.defaultdict(lambda: collections.defaultdict(list))
    per_shape = collections.defaultdict(lambda: collections.defaultdict(list))
    step_bytes, outside, n_steps = {}, 0.0, 1
    by_sym = collections.defaultdict(list)
    for e in (plan or []):
        by_sym[e[0]].append(tuple(e[1]))
    for cname, segs in segs_by_pass.items():
        n_steps = max(1, len(segs))
        hbm = cname in ('FETCH_SIZE', 'WRITE_SIZE')
        mult = 2048.0 if cname == 'FETCH_SIZE' else 1024.0
        if hbm:
            step_bytes[cname] = [sum(d[1].get(cname, 0.0) for d in sg) * mult for sg in segs]
            outside += sum(d[1].get(cname, 0.0) for d in all_by_pass[cname]) * mult - sum(step_bytes[cname])
        for sg in segs:
            seen = collections.Counter()
            for key, vals, dur in sg:
                j = seen[key]
                seen[key] += 1
                shape = by_sym[key][j] if j < len(by_sym.get(key, ())) else None
                for target in ([per_sym[key]] + ([per_shape[(key, shape)]] if shape else [])):
                    if hbm:
                        target[cname].append(vals.get(cname, 0.0) * mult)
                    else:
                        for c, v in vals.items():
                            target[c].append(v)
                        target['ns'].append(dur)

    def fold(t):
        mean = lambda v: sum(v) / len(v) if v else None      # noqa: E731
        o = {'launches_per_step': max((len(v) for v in t.values()), default=0) / n_steps}
        if t.get('FETCH_SIZE') and t.get('WRITE_SIZE'):
            o['bytes'] = mean(t['FETCH_SIZE']) + mean(t['WRITE_SIZE'])
        mf, ga = mean(t.get('SQ_VALU_MFMA_BUSY_CYCLES', [])), mean(t.get('GRBM_GUI_ACTIVE', []))
        if mf is not None and ga:
            # SQ_VALU_MFMA_BUSY_CYCLES: busy cycles of the matrix pipe summed over the 1024 SIMDs; GRBM_GUI_ACTIVE: active cycles summed over the 8 XCDs
            o['mfma_busy'] = mf / (N_SIMD * ga / 8.0)
            ns = mean(t.get('ns', []))
            o['clock_GHz'] = (ga / 8.0 / ns) if ns else None
        return o
    per_step = [f + w for f, w in zip(step_bytes['FETCH_SIZE'], step_bytes['WRITE_SIZE'])] if len(step_bytes) == 2 else []
    return {'kernels': {k: fold(v) for k, v in per_sym.items()}, 'shapes': {k: fold(v) for k, v in per_shape.items()},
            'bytes_per_step': (sum(per_step) / len(per_step)) if per_step else None, 'bytes_per_step_each': per_step, 'bytes_outside_the_steps': outside}


def measure_pmc(args, timeout_s=170, inference=False):
    """Hardware counters of this very command, measured now: child runs of this script under `rocprofv3 --pmc` -- FETCH_SIZE, WRITE_SIZE (separate
    passes, FETCH_SIZE doubled for gfx950: MI355X_MICROARCH.md) and SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE -- over 1 warm-up + 2 timed steps
    with no instrumentation of its own.  inference=True: the same three passes over the inference plan of the x3 and the bf16 mode (the fused
    feed-forward block of `roofline_ffn`).  Returns None when rocprofv3 is missing or a pass fails (the line then says so)."""
    import shutil, tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix='hftt_pmc_', dir='/tmp')
    plan_file = os.path.join(tmp, 'plan.json')
    child = [sys.executable if os.path.basename(sys.executable).startswith('python') else 'python3', os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--batch', str(args.batch),
             '--config', args.config, '--precision', args.precision, '--dropout', str(args.dropout), '--data', args.data,
             '--no-cpu-baseline', '--no-profile', '--no-extras', '--no-pmc'] + (['--inference-only'] if inference else [])
    env = dict(os.environ, TMPDIR='/tmp', HFTT_BENCH_PLAN_DUMP=plan_file)
    env.pop('RANK', None); env.pop('WORLD_SIZE', None); env.pop('LOCAL_RANK', None)
    try:
        passes = {}
        for tag, counters in (('FETCH_SIZE', ['FETCH_SIZE']), ('WRITE_SIZE', ['WRITE_SIZE']), ('busy', ['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'])):
            passes[tag] = _pmc_pass(exe, counters, child, env, tmp, tag, timeout_s)
            if passes[tag] is None and tag != 'busy':
                return None
        plans = json.load(open(plan_file)) if os.path.exists(plan_file) else {}
        passes = {k: v for k, v in passes.items() if v is not None}
        if inference:
            # the child runs three forwards of the inference plan in the x3 mode, then three in the bf16 mode: six im2win-delimited segments,
            # of which the last two of each mode are summarised against that mode's own plan
            res = {}
            cut = {k: _segments(v, 'im2win_kernel', False) for k, v in passes.items()}
            if any(len(v) != 6 for v in cut.values()):
                return None
            for mode, (lo, hi) in (('x3', (1, 3)), ('bf16', (4, 6))):
                res[mode] = _summarise({k: v[lo:hi] for k, v in cut.items()}, passes, plans.get('inference:%s' % mode))
            res['source'] = ('this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE child passes of `bench.py --inference-only` '
                             '(inference plan, three forwards in the x3 mode, then three in the bf16 mode; the last two of each; launches told apart by shape through the plan order)')
            return res
        cut = {k: _segments(v, 'adam_kernel', True) for k, v in passes.items()}
        if any(len(v) < 2 for v in cut.values()):
            return None
        out = _summarise({k: v[-2:] for k, v in cut.items()}, passes, plans.get('train:%s' % args.precision))
        out['divisor'] = 'dispatches between two adam_kernel launches = one step; the last two of the child\'s three steps, averaged'
        out['source'] = ('this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE child passes of the same command '
                         '(1 warm-up + 2 steps; only dispatches inside the two timed steps are counted)')
        return out
    except Exception as e:      # noqa: BLE001  (a failed counter pass must not take the bench line down)
        sys.stderr.write('measure_pmc failed: %r\n' % (e,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_object(key, v, steps, peak_tf, total_ms, kind='bench', mfma_passes=1, shape=None, live_mode=None):
    """roofline object of one kernel symbol from the profiling pass: ALGORITHMIC flops / bytes of its launches (engine plan meta, DESIGN.md
    section 5) over the measured launch durations; the bound is the side of the ridge its arithmetic intensity falls on."""
    ai = v['flops'] / max(v['bytes'], 1.0)
    ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
    tf = v['flops'] / (v['ms'] * 1e-3) / 1e12
    gbs = v['bytes'] / (v['ms'] * 1e-3) / 1e9
    if ai >= ridge:
        roof = {'bound': 'mfma', 'achieved': tf, 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': tf / peak_tf}
    else:
        roof = {'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS}
    # Counters: THIS run's (measure_pmc: child passes of the same command, dispatches inside the timed steps only), per (symbol, shape) when
    # the object is about one shape; the newest committed profile of the same configuration with the same symbol only as the stated fallback.
    live = (_LIVE_PMC if kind == 'bench' else (_LIVE_PMC_INF or {}).get(live_mode)) if (_LIVE_PMC if kind == 'bench' else _LIVE_PMC_INF) else None
    traffic = src = busy = clock = None
    if live is not None:
        ent = live['shapes'].get((key, tuple(shape))) if shape is not None else None
        ent = ent or live['kernels'].get(key)
        if ent is not None:
            traffic, busy, clock = ent.get('bytes'), ent.get('mfma_busy'), ent.get('clock_GHz')
            src = (_LIVE_PMC if kind == 'bench' else _LIVE_PMC_INF)['source']
    if traffic is None:
        traffic, src = pmc_traffic_bytes(key, kind)
        if src is not None:
            src = 'committed profile (no live counters in this run): ' + src
    if busy is None:
        busy = pmc_busy(key, kind)
    alg = v['bytes'] / v['launches']
    rejected = None
    if traffic is not None and traffic < 0.98 * alg:
        # fewer fabric bytes than the operands and results hold once cannot be this launch's HBM traffic (a stale or mis-keyed figure, or
        # lines served by a cache level the counter does not see): not printed as `traffic`
        rejected, traffic = traffic, None
    # two unambiguous fractions of the dense bf16 MFMA peak, whatever `bound` says: the ALGORITHMIC one (one multiply-add per product of the
    # reference's arithmetic -- what north_star's ">= 40 % on the FFN GEMMs" is priced in) and the matrix pipe's own (the split-operand mode
    # executes `mfma_passes` bf16-rate passes per product, so the pipe is mfma_passes x as busy as the algorithmic figure says)
    roof.update({'mfma_passes': mfma_passes, 'frac_of_bf16_mfma_peak': tf / PEAK_BF16_TFLOPS, 'matrix_pipe_tflops': mfma_passes * tf,
                 'matrix_pipe_frac_of_bf16_peak': mfma_passes * tf / PEAK_BF16_TFLOPS,
                 # context, not the contract's peak: what a bare back-to-back v_mfma_f32_32x32x16_bf16 loop delivers on this chip (32.0 cycles per
                 # instruction on all 1024 SIMDs at the 1.85 GHz the power management grants it on varied operands, steady state: tools/probes/mfma_chain.hip, profiles/r05_fp8_cross_terms.txt)
                 'matrix_pipe_frac_of_measured_mfma_loop': mfma_passes * tf / MEASURED_BF16_MFMA_LOOP_TFLOPS})
    roof.update({'traffic': traffic, 'traffic_source': src, 'traffic_rejected_below_algorithmic_bytes': rejected, 'mfma_busy': busy, 'clock_GHz': clock, 'kernel': key,
                 'launches_per_step': v['launches'] / steps, 'avg_launch_ms': v['ms'] / v['launches'],
                 'share_of_step_device_time': v['ms'] / max(total_ms, 1e-9), 'arithmetic_intensity': ai,
                 'algorithmic_flops_per_launch': v['flops'] / v['launches'], 'algorithmic_bytes_per_launch': v['bytes'] / v['launches'],
                 'tflops': tf, 'gbs': gbs, 'host_stall_intervals_replaced': v['stalls']})
    return roof


def store_batches(cfg, B, n_batches, rank, world, dev):
    """--data store: clips drawn from a MAESTRO-format store resident in HBM (training/dataset.py DeviceClipStore over
    corpus/make_dataset.synth_store, the clip contract pinned against the reference's MyDataset in tests/test_dataset_checkpoint.py)."""
    from corpus.make_dataset import synth_store
    from training.dataset import MyDataset, DeviceClipStore
    conf = {'feature': {'mel_bins': cfg.n_bin, 'n_bins': cfg.n_bin, 'log_offset': 1e-8},
            'input': {'margin_b': cfg.n_margin, 'margin_f': cfg.n_margin, 'num_frame': cfg.n_frame},
            'midi': {'num_note': cfg.n_note, 'num_velocity': cfg.n_velocity}}
    need = B * n_batches * world
    store = synth_store(conf, [need * 8 + 17, need * 8 + 5], seed=1234)          # two "files"; n_slice 16 below keeps >= need clips
    ds = MyDataset.from_arrays(store['feature'], store['label_onset'], store['label_offset'], store['label_mpe'], store['label_velocity'],
                               store['idx'], conf, 16)
    clips = DeviceClipStore(ds, dev)
    loader = clips.loader(B, rank=rank, world=world, shuffle=True, seed=1234, drop_last=True)
    chunks = loader.chunks[:n_batches]
    return clips, chunks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='clips per GPU')
    ap.add_argument('--config', default='paper', choices=['paper', 'tiny'])
    ap.add_argument('--precision', default='x3', choices=['x3', 'bf16', 'parity'])
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--data', default='synthetic', choices=['synthetic', 'store'],
                    help="'store': every step gathers its clips from a device-resident MAESTRO-format store (the gather is inside the step)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='skip the per-launch HIP-event pass (roofline objects become null)')
    ap.add_argument('--no-extras', action='store_true', help='skip the inference / parity-mode / compatibility-path legs')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 --pmc child passes (HBM traffic and MFMA-busy of this run; N = 1 only)')
    ap.add_argument('--inference-only', action='store_true',
                    help='(the command measure_pmc wraps for the inference plan) three eval forwards in the x3 mode, then three in the bf16 mode; no JSON line')
    args = ap.parse_args()

    import gc
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
    # HFTT_BENCH_FORCE_DDP=1 (rehearsal on a one-GPU box): take the N > 1 code path with a one-rank RCCL group -- init_process_group('nccl'),
    # parameter broadcast, the bucketed all-reduce on the side stream, the barriers -- so that branch has run on real hardware before the
    # multi-GPU node sees it
    force_ddp = os.environ.get('HFTT_BENCH_FORCE_DDP') == '1'
    if world > 1 or force_ddp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    collective = None
    if world > 1 or force_ddp:
        # what the collective really saw: every rank reports (rank, local rank, device index, device name, uuid) through an object all-gather
        props = torch.cuda.get_device_properties(dev)
        mine = {'rank': rank, 'local_rank': local_rank, 'device': dev_index, 'name': props.name, 'uuid': str(getattr(props, 'uuid', ''))}
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, mine)
        collective = {'backend': dist.get_backend(), 'ranks': dist.get_world_size(), 'devices': seen}

    cfg = CONFIGS[args.config]
    global _PROFILE_CONFIG, _PROFILE_PRECISION
    _PROFILE_CONFIG, _PROFILE_PRECISION = args.config, args.precision
    B = args.batch
    model = build_model(cfg, 1234, args.dropout, dev)
    model.hftt_precision = args.precision
    model.train()
    plan_dump = os.environ.get('HFTT_BENCH_PLAN_DUMP')
    if args.inference_only:
        model.eval()
        model.hftt_freeze_weights(True)              # what model.amt.AMT does: a transcriber's weights are constant
        xs = [synthetic_batch(cfg, B, 1234 + i, dev)[0] for i in range(2)]
        with torch.no_grad():
            for mode in ('x3', 'bf16'):
                model.hftt_precision = mode
                model.hftt_freeze_weights(True)
                for i in range(3):
                    model(xs[i % 2])
                torch.cuda.synchronize()
                if plan_dump:
                    dump_plans(model, B, plan_dump, ['inference'])
        return
    grad_sync = None
    if world > 1 or force_ddp:
        from hftt_hip.ddp import FlatGradSync, broadcast_parameters
        eng = model.hftt_engine()
        broadcast_parameters(eng)
        if force_ddp and world == 1:
            dist.broadcast(eng.flat_params, 0)                 # (broadcast_parameters skips a one-rank job)
        grad_sync = FlatGradSync(eng, world, rank=rank)        # folds the rank into the dropout seed: every rank draws its own masks
    ts = TrainStep(model, lr=1e-4, grad_sync=grad_sync)

    # clips resident in HBM before the timed region (per-rank shard): random tensors of the clip contract, or a device-resident store
    n_batches = 4
    if args.data == 'store':
        clips, chunks = store_batches(cfg, B, n_batches, rank, world, dev)

        def batch(i):
            b = clips.batch(chunks[i % n_batches])
            return b[0], b[1:]
    else:
        data = [synthetic_batch(cfg, B, 1234 + 1000 * rank + i, dev) for i in range(n_batches)]

        def batch(i):
            return data[i % n_batches]

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: exactly `steps` training steps, no profiler, no host sync inside
    for i in range(args.warmup):
        x, lab = batch(i)
        ts(x, *lab)
    gc.collect()
    gc.disable()
    # one event per timed step on the stream the step is enqueued on (recorded, never waited for inside the region): the dispersion of the
    # steps the one number below is made of
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        x, lab = batch(i)
        loss = ts(x, *lab)
    marks[args.steps].record()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    loss_val = float(loss[0].item())
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if plan_dump:
        dump_plans(model, B, plan_dump, ['train'])
    if world > 1 or force_ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if share:
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def timed(fn, n, warm=1):
        for _ in range(warm):
            fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # ---- profiling pass (after the timed region): per-launch HIP events, harvested after every step.  EVERY rank runs it (the step
    # contains the gradient all-reduce: a pass on rank 0 alone would wait for its peers forever); rank 0 reports.
    prof = None
    psteps = max(2, min(5, args.steps))
    if not args.no_profile:
        prof = LaunchProfiler()
        ts.engine.profiler = prof
        for i in range(psteps):
            x, lab = batch(i)
            ts(x, *lab)
            prof.step_done()
        ts.engine.profiler = None
        sync()

    overlap = None
    if grad_sync is not None and dev.type == 'cuda':
        # two more steps with an event behind every bucket's all-reduce and one at the end of the backward: did the early buckets (time
        # decoder, frequency decoder) finish under the encoder's backward, as the design says?  (every rank runs them: they all-reduce)
        grad_sync.timing = True
        for i in range(2):
            x, lab = batch(i)
            ts(x, *lab)
        sync()
        overlap = grad_sync.overlap_report()
        grad_sync.timing = False

    if rank == 0 and world == 1 and not force_ddp and not args.no_pmc and not args.no_profile:
        global _LIVE_PMC, _LIVE_PMC_INF
        _LIVE_PMC = measure_pmc(args)
        if not args.no_extras and args.config == 'paper':
            _LIVE_PMC_INF = measure_pmc(args, inference=True)

    result = None
    if rank == 0:
        value = B * world * args.steps / dt
        peak_tf = PEAK_F32_TFLOPS if args.precision == 'parity' else PEAK_BF16_TFLOPS
        passes = 3 if args.precision == 'x3' else 1
        roof = roof_ffn = kernels = None
        extras = {}
        if prof is not None:
            summ = prof.summary()
            total_ms = sum(v['ms'] for v in summ.values())
            # dominant kernel among the calls that launch exactly one kernel (hftt_gemm_tn launches its split kernel plus a slab reduce, so
            # its event interval is not one kernel's duration and would not match rocprofv3's per-kernel average)
            single = {k: v for k, v in summ.items() if not k.startswith('gemm_tn') and v['flops'] > 0}
            key, dom = max(single.items(), key=lambda kv: kv[1]['ms'])
            roof = roofline_object(key, dom, psteps, peak_tf, total_ms, mfma_passes=passes)
            # continuity: rounds 3-5 and the first builds of round 6 had the attention backward as the dominant call; since the attention output
            # projection + LayerNorm + FFN became one launch (9 per step) that launch is -- the attention backward's object rides along
            ab = {k: v for k, v in single.items() if 'attn_bwd' in k and k != key}
            if ab:
                k2, v2 = max(ab.items(), key=lambda kv: kv[1]['ms'])
                extras['roofline_attention_backward'] = roofline_object(k2, v2, psteps, peak_tf, total_ms, mfma_passes=passes)
            top = sorted(summ.items(), key=lambda kv: -kv[1]['ms'])[:14]
            kernels = [{'kernel': k, 'ms_per_step': v['ms'] / psteps, 'launches_per_step': v['launches'] / psteps,
                        'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['flops'] else None,
                        'gbs': (v['bytes'] / (v['ms'] * 1e-3) / 1e9) if v['bytes'] else None} for k, v in top]
            extras['profiled_device_ms_per_step'] = total_ms / psteps
            # the same kernels split by problem shape (one symbol serves the encoder's S_e = B*T*F tokens and the decoder's S_n = B*T*N)
            by_shape = {}
            for k, v in prof.samples.items():
                for m_, t_ in zip(v['meta'], v['ms']):
                    e = by_shape.setdefault((k, tuple(m_.get('shape', ()))), [0, 0.0, 0.0, 0.0])
                    e[0] += 1; e[1] += t_; e[2] += m_['flops']; e[3] += m_['bytes']
            big = sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:16]
            extras['kernels_by_shape'] = [{'kernel': k, 'shape': list(sh), 'launches_per_step': e[0] / psteps, 'avg_launch_us': 1e3 * e[1] / e[0],
                                           'tflops': e[2] / (e[1] * 1e-3) / 1e12 if e[2] else None, 'gbs': e[3] / (e[1] * 1e-3) / 1e9 if e[3] else None}
                                          for (k, sh), e in big]
        if world == 1 and not args.no_extras:
            x0, lab0 = batch(0)

            def inference_leg(precision):
                """inference plan (model.eval(): nothing saved for a backward) at the same batch: clips/s, the six posteriors of batch 0 and the
                roofline object of the fused feed-forward block hftt_ffn_res_ln_fwd at S_e tokens"""
                model.hftt_precision = precision
                model.eval()
                model.hftt_freeze_weights(True)            # what model.amt.AMT does: a transcriber's weights are constant
                out = {}
                with torch.no_grad():
                    t_inf = timed(lambda i: model(batch(i)[0]), 10, warm=2)
                    out['post'] = [t.clone() for k, t in enumerate(model(x0)) if k in (0, 1, 2, 5, 6, 7)]
                    out['logits'] = [t.clone() for k, t in enumerate(model(x0)) if k in (3, 8)]
                    out['clips_per_s'] = B / t_inf
                    if not args.no_profile and precision != 'parity':
                        pr = LaunchProfiler()
                        eng = model.hftt_engine()
                        eng.profiler = pr
                        for i in range(3):
                            model(batch(i)[0])
                            pr.step_done()
                        eng.profiler = None
                        s_inf = pr.summary()
                        tot_inf = sum(v['ms'] for v in s_inf.values())
                        ffn = {k: v for k, v in s_inf.items() if ('_mlp' in k) and v['meta'] and v['meta'][0]['shape'][0] >= 200000}
                        if ffn:
                            k_ffn, v_ffn = max(ffn.items(), key=lambda kv: kv[1]['ms'])
                            sel = [(m_, t_) for m_, t_ in zip(v_ffn['meta'], pr.samples[k_ffn]['ms']) if m_['shape'][0] >= 200000]
                            v_sel = {'launches': len(sel), 'ms': sum(t_ for _, t_ in sel), 'flops': sum(m_['flops'] for m_, _ in sel),
                                     'bytes': sum(m_['bytes'] for m_, _ in sel), 'stalls': 0}
                            rf = roofline_object(k_ffn, v_sel, 3, PEAK_BF16_TFLOPS, tot_inf, kind='inference', mfma_passes=3 if precision == 'x3' else 1,
                                                 shape=sel[0][0]['shape'], live_mode=precision)
                            fused = bool(sel[0][0].get('fused'))       # round 6: fc_o + residual + LayerNorm + FFN as one launch (flops and bytes of both halves)
                            rf['entry_point'] = 'hftt_attn_out_ffn_fwd' if fused else 'hftt_ffn_res_ln_fwd'
                            rf['precision_mode'] = precision
                            rf['plan'] = 'inference (no hidden / pre-LN stores%s), tokens per launch %d' % (
                                '; the attention output projection + LayerNorm in front of the block in the same launch, its output never written' if fused else '',
                                sel[0][0]['shape'][0])
                            out['roofline_ffn'] = rf
                model.hftt_freeze_weights(False)
                return out

            def train_leg(precision):
                model.hftt_precision = precision
                model.train()
                ts_m = TrainStep(model, lr=1e-4, optimizer=ts.opt)
                t_m = timed(lambda i: ts_m(batch(i)[0], *batch(i)[1]), 3 if precision == 'parity' else 6, warm=1)
                del ts_m
                return B / t_m

            inf = {args.precision: inference_leg(args.precision)}
            extras['inference_clips_per_s'] = inf[args.precision]['clips_per_s']
            roof_ffn = inf[args.precision].get('roofline_ffn')
            others = [m for m in ('x3', 'bf16', 'parity') if m != args.precision and not (args.config != 'paper' and m == 'parity')]
            for m in others:
                inf[m] = inference_leg(m)

            def diff(a, b, what):
                return max(float((x - y).abs().max()) for x, y in zip(inf[a][what], inf[b][what]))
            for m in others:
                key = {'bf16': 'bf16_mode', 'x3': 'x3_mode', 'parity': 'fp32_mfma_mode'}[m]
                extras[key] = {'clips_per_s': train_leg(m), 'inference_clips_per_s': inf[m]['clips_per_s'],
                               'max_abs_diff_posteriors_vs_benchmarked_mode': diff(m, args.precision, 'post'),
                               'max_abs_diff_velocity_logits_vs_benchmarked_mode': diff(m, args.precision, 'logits'),
                               'what': 'training step / eval forward in precision mode "%s" on the same clips; differences of the six posteriors and '
                                       'the two velocity-logit tensors against the benchmarked mode "%s" (eval forward)' % (m, args.precision)}
                if m == 'bf16':
                    # the single-pass mode is outside north_star's 1e-3 and, on harmonic input, is not a training mode
                    # (gradient cosine 0.09 - 0.25 against the exact mode): its figures are quoted for inference only
                    extras[key]['scope'] = ('inference only: single-pass bf16 operands are outside the 1e-3 output budget (the differences above) and the mode '
                                            'carries no training-quality claim; clips_per_s is the time of its training step, not a training result')
                if 'roofline_ffn' in inf[m]:
                    extras[key]['roofline_ffn'] = inf[m]['roofline_ffn']
            from hftt_hip import _capi as _hc
            if args.precision == 'x3' and args.config == 'paper' and (_hc.lib().hftt_build_options() & 1):      # (only a HFTT_BUILD_GRAD_HI=1 library has it)
                # the opt-in of DESIGN.md section 3: gradient operands of the backward GEMMs as their bf16 rounding (two MFMA passes)
                os.environ['HFTT_X3_GRAD_HI'] = '1'
                model.hftt_precision = 'bf16'; model.hftt_engine()          # (the engine reads the switch when its precision is set)
                try:
                    extras['x3_gradient_rounding_option'] = {'clips_per_s': train_leg('x3'), 'env': 'HFTT_X3_GRAD_HI=1', 'default': False,
                                                             'what': 'x3 with the gradient operand of every backward GEMM rounded to bf16 (same forward: same outputs)'}
                finally:
                    del os.environ['HFTT_X3_GRAD_HI']
                    model.hftt_precision = 'bf16'; model.hftt_engine()
            model.hftt_precision = args.precision
            # ---- front end (model/amt.py:55-63) on config 5's minute: 60 s of 44.1 kHz mono -> hftt_resample -> 16 kHz -> hftt_logmel -> [3751, 256]
            try:
                from hftt_hip import ops as _ops
                g_fe = torch.Generator().manual_seed(1234)
                wav = (0.1 * torch.randn(60 * 44100, generator=g_fe)).to(dev)
                lm = _ops.LogMel(dev)
                rounds = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(6)]
                for e in rounds:                                         # (the fastest of the last five rounds is reported: the intervals include the
                    e[0].record(); w16 = _ops.resample(wav, 44100, 16000)   # host side of the two calls, and one preempted host thread is a 40 ms outlier)
                    e[1].record(); feat = lm(w16)
                    e[2].record()
                torch.cuda.synchronize()
                t_rs = min(e[0].elapsed_time(e[1]) for e in rounds[1:]) * 1e-3
                t_lm = min(e[1].elapsed_time(e[2]) for e in rounds[1:]) * 1e-3
                b_rs, b_lm = 4.0 * (wav.numel() + w16.numel()), 4.0 * (w16.numel() + feat.numel())
                extras['front_end'] = {'workload': '60 s of 44.1 kHz mono audio -> polyphase resample to 16 kHz -> 2048-point STFT, 256 htk mels (slaney), log: %d frames' % feat.shape[0],
                                       'resample_s': t_rs, 'logmel_s': t_lm, 'resample_GBps': b_rs / t_rs / 1e9, 'logmel_GBps': b_lm / t_lm / 1e9,
                                       'audio_seconds_per_second': 60.0 / (t_rs + t_lm), 'algorithmic_bytes': {'resample': b_rs, 'logmel': b_lm},
                                       'note': 'one minute is 10.6 MB in and 3.8 MB out: these launches are latency-, not bandwidth-bound; includes the host side of the two calls'}
            except Exception as ex:      # noqa: BLE001
                extras['front_end'] = {'error': repr(ex)}
            # ---- compatibility path: the reference's loop unchanged (torch.optim.Adam, 8 nn criteria, loss.backward()) through training.train
            import torch.nn as nn
            from training import train as T
            model.train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-4)
            crits = [nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss(), nn.BCELoss(), nn.BCELoss(), nn.BCELoss(), nn.CrossEntropyLoss()]
            it = [(batch(i)[0],) + tuple(batch(i)[1]) for i in range(n_batches)]
            T.train(model, it[:1], opt, *crits, 1.0, 1.0, dev, False)
            torch.cuda.synchronize()
            t0c = time.perf_counter()
            T.train(model, it, opt, *crits, 1.0, 1.0, dev, False)
            torch.cuda.synchronize()
            extras['compat_path_clips_per_s'] = B * n_batches / (time.perf_counter() - t0c)
        result = {
            'metric': 'training clips/sec (128-frame x 256-bin)', 'value': value, 'unit': 'clips/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'x3': 'f16x3 (split fp16 / bf16 operand pairs, three bf16-rate MFMA passes per product, fp32 accumulate; tensors that flow between '
                            'kernels are fp32 (attention q/k/v as fp16 hi+lo pairs), the copies SAVED only for the backward -- FFN hidden, its gradient, '
                            'pre-LayerNorm sums -- are bf16)', 'bf16': 'bf16', 'parity': 'f32'}[args.precision],
            'data': 'synthetic' if args.data == 'synthetic' else 'synthetic (MAESTRO-format store resident in HBM, clips gathered inside the step)',
            'config': {'workload': '%s-size hFT-Transformer training step (d=%d, ff=%d, %d+%d layers, %d heads), batch %d clips/GPU, '
                                   'dropout %.2f, forward+loss+backward+Adam' % (args.config, cfg.hid_dim, cfg.pf_dim, cfg.enc_layer, cfg.dec_layer,
                                                                                cfg.enc_head, B, args.dropout),
                       'global_batch': B * world, 'frames': cfg.n_frame, 'bins': cfg.n_bin, 'parallelism': 'dp%d' % world,
                       'precision_mode': args.precision},
            'step_ms_min': step_ms[0], 'step_ms_median': step_ms[len(step_ms) // 2], 'step_ms_max': step_ms[-1],
            'step_ms_source': 'one HIP event per timed step on the compute stream (elapsed between consecutive events), %d steps' % args.steps,
            'model_tflops': 3 * FWD_GFLOP_PER_CLIP * value / 1e3 if args.config == 'paper' else None,
            'final_loss': loss_val,
            'roofline': roof,
            'roofline_ffn': roof_ffn,
            'kernels': kernels,
            'collective': collective,
        }
        step_bytes, step_src = pmc_step_bytes()
        if _LIVE_PMC is not None and _LIVE_PMC.get('bytes_per_step'):
            step_bytes, step_src = _LIVE_PMC['bytes_per_step'], _LIVE_PMC['source']
            result['hbm_bytes_outside_the_steps'] = _LIVE_PMC.get('bytes_outside_the_steps')
            result['hbm_bytes_per_step_each'] = _LIVE_PMC.get('bytes_per_step_each')
            result['hbm_bytes_divisor'] = _LIVE_PMC.get('divisor')
        if step_bytes is not None:
            # whole-step HBM traffic (PMC sum over every kernel of a profiled run of this same command) over THIS run's step time
            result['hbm_bytes_per_step'] = step_bytes
            result['hbm_frac_of_peak'] = step_bytes / (dt / args.steps) / (PEAK_HBM_GBS * 1e9)
            result['hbm_bytes_source'] = step_src
        if collective is not None and overlap is not None:
            collective['buckets'] = overlap
            collective['early_buckets_hidden'] = all(b['ms_before_backward_end'] > 0.0 for b in overlap[:-1])
        result.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(cfg, config_name=args.config)
        else:
            result['cpu_baseline'] = None
        print(json.dumps(result))
    if world > 1 or force_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
