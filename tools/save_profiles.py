#!/usr/bin/env python3
"""Copy a tools/profile_bench.sh run out of gpurun_out/ into profiles/ (tracked).   Usage: python tools/save_profiles.py <tag>
Writes profiles/<tag>_bench.json (the bench line), _bench_kernel_stats.csv, _bench_pmc_traffic.json, _bench_pmc_busy.json and the same
three for the inference plan (<tag>_inference_*)."""
import collections
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
MODE = sys.argv[2] if len(sys.argv) > 2 else 'paper size, B=8, x3 mode'
# the BENCH_ARGS the profiled commands ran with (third argument, e.g. "--config tiny --precision bf16"): written into every summary as
# explicit fields -- bench.py picks the summaries of ITS configuration by these fields, not by the prose of `note`
BENCH_ARGS = sys.argv[3] if len(sys.argv) > 3 else ''
_m = re.search(r'--config\s+(\w+)', BENCH_ARGS)
CONFIG = _m.group(1) if _m else 'paper'
_m = re.search(r'--precision\s+(\w+)', BENCH_ARGS)
PRECISION = _m.group(1) if _m else 'x3'
FIELDS = {'config': CONFIG, 'precision': PRECISION, 'bench_args': BENCH_ARGS}
G = 'gpurun_out/' + tag


def key_of(n):
    m = re.search(r'::([a-z_0-9]+(?:<[^>]*>)?)\(', n)
    return m.group(1) if m else n


def load(path):
    """-> {kernel symbol: {counter: [values per launch]}} (rocprofv3 counter_collection.csv, one row per dispatch and counter)"""
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        if 'namespace)::' in n and 'at::' not in n:
            d[key_of(n)][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


def traffic(fdir, wdir, out, what):
    f, w = load(f'{fdir}/b_counter_collection.csv'), load(f'{wdir}/b_counter_collection.csv')
    k = {}
    for n in sorted(set(f) | set(w)):
        fl, wl = f[n].get('FETCH_SIZE', [0]), w[n].get('WRITE_SIZE', [0])
        k[n] = {'launches': len(fl), 'fetch_MB_per_launch_x2_corrected': round(2 * sum(fl) / len(fl) * 1024 / 1e6, 2),
                'write_MB_per_launch': round(sum(wl) / max(1, len(wl)) * 1024 / 1e6, 2)}
    m = re.search(r'--steps (\d+)', what)
    w_ = re.search(r'--warmup (\d+)', what)
    n_steps = (int(m.group(1)) if m else 2) + (int(w_.group(1)) if w_ else 0)       # every step of the profiled command (warm-up included) is in the counters
    total = sum(v['launches'] * (v['fetch_MB_per_launch_x2_corrected'] + v['write_MB_per_launch']) for v in k.values()) * 1e6
    json.dump(dict(FIELDS, **{'note': ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) around `%s` (%s); counter '
                        'values are KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated on a 268 MB copy: '
                        'profiles/r01_kernel_pmc_traffic.json); per-launch averages over all launches of a kernel symbol') % (what, MODE),
                        'steps_in_profiled_command': n_steps, 'hbm_bytes_per_step': total / n_steps, 'kernels': k}),
              open(out, 'w'), indent=1)


def durations(path):
    """kernel symbol -> mean duration (ns) of its dispatches in a pass's kernel trace"""
    d = collections.defaultdict(list)
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            n = r['Kernel_Name']
            if 'namespace)::' in n and 'at::' not in n:
                d[key_of(n)].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    return {k: sum(v) / len(v) for k, v in d.items()}


N_SIMD = 256 * 4


def busy(dirs, out, what):
    d = collections.defaultdict(dict)
    for p in dirs:
        dur = durations(f'{p}/b_kernel_trace.csv')
        for n, c in load(f'{p}/b_counter_collection.csv').items():
            for name, vals in c.items():
                d[n][name] = sum(vals) / len(vals)
                d[n]['launches'] = len(vals)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
                d[n]['duration_ns_in_this_pass'] = dur.get(n)
    k = {}
    for n, c in sorted(d.items()):
        e = {'launches': c.get('launches')}
        e.update({name: round(v, 1) for name, v in c.items() if name != 'launches' and v is not None})
        mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES')
        if mf is not None and c.get('GRBM_GUI_ACTIVE'):
            # SQ_VALU_MFMA_BUSY_CYCLES: cycles the MFMA pipe was busy, summed over all 1024 SIMDs (= 32 per v_mfma_f32_32x32x16_bf16);
            # GRBM_GUI_ACTIVE: active cycles summed over the 8 XCDs (MI355X_MICROARCH.md) -> per-SIMD busy fraction
            e['mfma_busy'] = round(mf / (N_SIMD * c['GRBM_GUI_ACTIVE'] / 8.0), 4)
            e['clock_GHz'] = round(c['GRBM_GUI_ACTIVE'] / 8.0 / c['duration_ns_in_this_pass'], 3) if c.get('duration_ns_in_this_pass') else None
        elif mf is not None and c.get('duration_ns_in_this_pass'):
            # no cycle count of the dispatch in this pass: price the duration at the 2.4 GHz peak clock (the chip runs ~1.9-2.0 GHz under a
            # profiled load, so this UNDERSTATES the busy fraction by up to ~20 %)
            e['mfma_busy'] = round(mf / (N_SIMD * c['duration_ns_in_this_pass'] * 2.4), 4)
            e['mfma_busy_basis'] = 'duration x 2.4 GHz (lower bound)'
        k[n] = e
    json.dump(dict(FIELDS, **{'note': 'rocprofv3 --pmc passes around `%s`; per-launch averages per kernel symbol. mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / '
                       '(1024 SIMDs x cycles of the dispatch); cycles of the dispatch = GRBM_GUI_ACTIVE / 8 where that counter was collected in the '
                       'same pass, else the traced duration at 2.4 GHz' % what, 'kernels': k}), open(out, 'w'), indent=1)


cmd = 'python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras'
shutil.copy(f'{G}_stats/bench_kernel_stats.csv', f'profiles/{tag}_bench_kernel_stats.csv')
if os.path.exists(f'{G}_bench.json'):
    shutil.copy(f'{G}_bench.json', f'profiles/{tag}_bench.json')
traffic(f'{G}_fetch', f'{G}_write', f'profiles/{tag}_bench_pmc_traffic.json', cmd)
busy([f'{G}_busy', f'{G}_busy2'], f'profiles/{tag}_bench_pmc_busy.json', cmd)
if os.path.exists(f'{G}_inf_stats/bench_kernel_stats.csv'):
    cmd = 'python3 tools/bench_inference.py --steps 2'
    shutil.copy(f'{G}_inf_stats/bench_kernel_stats.csv', f'profiles/{tag}_inference_kernel_stats.csv')
    traffic(f'{G}_inf_fetch', f'{G}_inf_write', f'profiles/{tag}_inference_pmc_traffic.json', cmd)
    busy([f'{G}_inf_busy'], f'profiles/{tag}_inference_pmc_busy.json', cmd)
for name in (f'profiles/{tag}_bench_kernel_stats.csv', f'profiles/{tag}_inference_kernel_stats.csv'):
    if os.path.exists(name):
        print(name)
        for r in list(csv.DictReader(open(name)))[:12]:
            print('  %-84s calls=%6s avg_us=%9.1f pct=%5s' % (r['Name'][:84], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
