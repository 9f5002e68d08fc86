#!/usr/bin/env python3
"""Idle time between consecutive kernels of the training step, from a rocprofv3 kernel trace of bench.py:
    cd /tmp && rocprofv3 --kernel-trace -d <dir> -o t --output-format csv -- python3 <repo>/bench.py --steps 3 --warmup 2 --no-extras --no-pmc --no-cpu-baseline --no-profile
    python tools/launch_gaps.py <dir>
A step = the dispatches between two adam_kernel launches (as bench.py::measure_pmc cuts them).  Prints, for the last step: kernels, busy time,
the sum of the gaps (start of kernel i+1 minus end of kernel i, when positive) and their distribution."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
f = glob.glob(os.path.join(sys.argv[1], '**', 't_kernel_trace.csv'), recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), bench._kernel_key(r['Kernel_Name'])) for r in csv.DictReader(open(f))), key=lambda r: r[0])
cuts = [i for i, r in enumerate(rows) if r[2] == 'adam_kernel']
a, b = cuts[-2] + 1, cuts[-1] + 1
step = rows[a:b]
busy = sum(e - s for s, e, _ in step)
gaps = [step[i + 1][0] - step[i][1] for i in range(len(step) - 1)]
pos = [g for g in gaps if g > 0]
span = step[-1][1] - step[0][0]
print('last step: %d kernels, span %.3f ms, busy %.3f ms, gaps %.3f ms (%.1f %% of the span), overlapped pairs %d' % (len(step), span / 1e6, busy / 1e6, sum(pos) / 1e6, 100.0 * sum(pos) / span, sum(1 for g in gaps if g <= 0)))
pos.sort()
if pos:
    print('gap ns: median %d, p90 %d, max %d; gaps > 10 us: %d' % (pos[len(pos) // 2], pos[int(0.9 * len(pos))], pos[-1], sum(1 for g in pos if g > 10000)))
big = sorted(((step[i + 1][0] - step[i][1], step[i][2], step[i + 1][2]) for i in range(len(step) - 1)), reverse=True)[:8]
for g, k0, k1 in big:
    print('  %7.1f us between %-44s and %s' % (g / 1e3, k0[:44], k1[:44]))
