#!/usr/bin/env python3
"""dev: the shader clock of every kernel of a training step.  A sampler (tools/probes/clock_probe.hip, built by tools/clock_trace.sh) is enqueued
on the step's stream in front of and behind every plan entry of the engine (the hook bench.py's LaunchProfiler uses): per XCD it records the
constant 100 MHz counter and the shader-clock counter; between two samples of the same XCD the ratio of the increments is the clock the
chip's power management granted that launch.  Prints one JSON object: per kernel launches, time, clock; and the time-weighted clock of the step."""
import argparse, ctypes as C, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import bench
ap = argparse.ArgumentParser()
ap.add_argument('--config', default='paper'); ap.add_argument('--precision', default='x3'); ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--out', default=None)
args = ap.parse_args()
from hftt_hip.trainer import TrainStep
dev = torch.device('cuda:0')
cfg = bench.CONFIGS[args.config]
model = bench.build_model(cfg, 1234, 0.1, dev); model.hftt_precision = args.precision; model.train()
ts = TrainStep(model, lr=1e-4)
data = [bench.synthetic_batch(cfg, 8, 1234 + i, dev) for i in range(2)]
for i in range(3):
    x, lab = data[i % 2]; ts(x, *lab)
torch.cuda.synchronize()
probe = C.CDLL(os.environ.get('CLOCK_PROBE_LIB', '/tmp/libclockprobe.so'))
probe.clock_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
CAP = 8192
WINDOW = 20                                                      # markers: 0.2 us
buf = torch.zeros(CAP * 3, dtype=torch.int64, device=dev)


class ClockSampler:
    """the interface of hftt_hip.profiler.LaunchProfiler: begin(name, meta) / end() around every plan entry; one sample behind each entry"""
    def __init__(self):
        self.n = 0; self.keys = []
    def begin(self, name, meta):
        self.keys.append(meta['kernel'] if meta else name)
    def end(self):
        assert self.n < CAP
        assert probe.clock_sample(buf.data_ptr(), self.n, WINDOW, torch.cuda.current_stream(dev).cuda_stream) == 0
        self.n += 1


cs = ClockSampler()
# the resident sampler: (realtime, shader counter) every 10 us on a side stream, long enough for the steps, then it leaves by itself
probe.clock_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
PERIOD = 1000
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for i in range(args.steps):
    x, lab = data[i % 2]; ts(x, *lab)
b.record(); torch.cuda.synchronize()
ms_plain = a.elapsed_time(b) / args.steps
NS = int(ms_plain * args.steps * 1.3 * 1e5 / PERIOD) + 200
trace = torch.zeros(NS * 2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream(dev)
assert probe.clock_trace(trace.data_ptr(), NS, PERIOD, side.cuda_stream) == 0
ts.engine.profiler = cs
a.record()
for i in range(args.steps):
    x, lab = data[i % 2]; ts(x, *lab)
b.record()
ts.engine.profiler = None
torch.cuda.synchronize()
t = buf.view(CAP, 3)[:cs.n].cpu().double()
tr = trace.view(NS, 2).cpu().double()
tr = tr[tr[:, 0] > 0]
mid = (tr[1:, 0] + tr[:-1, 0]) / 2                               # clock(t) between consecutive trace samples
clk = (tr[1:, 1] - tr[:-1, 1]) / (tr[1:, 0] - tr[:-1, 0]) * 0.1
import bisect
mids = mid.tolist(); clks = clk.tolist()
per = {}
for j, key in enumerate(cs.keys):
    if j == 0:
        continue
    t0 = (t[j - 1, 0] + t[j - 1, 1]).item(); t1 = t[j, 0].item()   # entry j ran between the end of marker j - 1 and the start of marker j
    if t1 <= t0 or t1 - t0 > 5e5:
        continue
    lo = bisect.bisect_left(mids, t0); hi = bisect.bisect_right(mids, t1)
    d = per.setdefault(key, {'launches': 0, 'rt': 0.0, 'cs': 0.0, 'cn': 0})
    d['launches'] += 1; d['rt'] += t1 - t0
    if hi > lo:
        d['cs'] += sum(clks[lo:hi]); d['cn'] += hi - lo
tot_rt = sum(v['rt'] for v in per.values())
rows = sorted(per.items(), key=lambda kv: -kv[1]['rt'])
inside = [c for m, c in zip(mids, clks) if t[0, 0].item() <= m <= t[cs.n - 1, 0].item()]
res = {'config': args.config, 'precision': args.precision, 'steps': args.steps,
       'method': 'a resident one-wave sampler on a side stream appends (s_memrealtime 100 MHz, s_memtime shader counter) every 10 us: clock(t) of its XCD while the step runs beside it; one-wave markers behind every plan entry give each launch\'s interval in the same 100 MHz time base; a kernel\'s clock = mean of the trace samples inside its intervals',
       'ms_per_step': ms_plain, 'ms_per_step_with_sampler_and_markers': a.elapsed_time(b) / args.steps,
       'clock_GHz_over_the_steps': sum(inside) / max(1, len(inside)), 'clock_GHz_min_10us': min(inside) if inside else None, 'clock_GHz_max_10us': max(inside) if inside else None,
       'kernels': [{'kernel': k, 'launches_per_step': round(v['launches'] / args.steps, 2), 'avg_us': round(v['rt'] * 1e-2 / v['launches'], 1),
                    'clock_GHz': (round(v['cs'] / v['cn'], 3) if v['cn'] else None), 'samples': v['cn'], 'share': round(v['rt'] / tot_rt, 4)} for k, v in rows[:40]]}
print(json.dumps(res))
if args.out:
    open(args.out, 'w').write(json.dumps(res, indent=1) + '\n')
