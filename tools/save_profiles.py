#!/usr/bin/env python3
"""Copy a rocprofv3 run out of gpurun_out/ into profiles/ (tracked).
Usage: python tools/save_profiles.py <tag> <stats_dir> <bench_json> <pmc_fetch_dir> <pmc_write_dir>"""
import csv, collections, json, re, shutil, sys
tag, stats_dir, bench_json, fdir, wdir = sys.argv[1:6]
shutil.copy(f'{stats_dir}/bench_kernel_stats.csv', f'profiles/{tag}_bench_kernel_stats.csv')
shutil.copy(bench_json, f'profiles/{tag}_bench.json')
def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r['Kernel_Name']].append(float(r['Counter_Value']))
    return d
f = load(f'{fdir}/b_counter_collection.csv'); w = load(f'{wdir}/b_counter_collection.csv')
out = {}
for n in sorted(set(f) | set(w)):
    if 'namespace)::' in n and 'at::' not in n:
        m = re.search(r'::([a-z_0-9]+(?:<[^>]*>)?)\(', n)
        key = m.group(1) if m else n
        fl = f.get(n, [0]); wl = w.get(n, [0])
        out[key] = {'launches': len(fl), 'fetch_MB_per_launch_x2_corrected': round(2 * sum(fl) / len(fl) * 1024 / 1e6, 2),
                    'write_MB_per_launch': round(sum(wl) / max(1, len(wl)) * 1024 / 1e6, 2)}
json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) around `python3 bench.py --steps 2 --warmup 1 '
                   '--no-cpu-baseline --no-profile` (paper size, B=8, bf16 mode); counter values are KB; FETCH_SIZE doubled as '
                   'MI355X_MICROARCH.md prescribes for gfx950 (calibrated on a 268 MB copy: profiles/r01_kernel_pmc_traffic.json); '
                   'per-launch averages over all launches of a kernel symbol', 'kernels': out},
          open(f'profiles/{tag}_bench_pmc_traffic.json', 'w'), indent=1)
rows = list(csv.DictReader(open(f'profiles/{tag}_bench_kernel_stats.csv')))
for r in rows[:10]:
    print('%-84s calls=%6s avg_us=%9.1f pct=%5s' % (r['Name'][:84], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
