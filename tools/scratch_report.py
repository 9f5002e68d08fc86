#!/usr/bin/env python3
"""Per-kernel register / scratch usage of every HIP source of the default library (hipcc -Rpass-analysis=kernel-resource-usage with
build.py's flags).  Usage: python tools/scratch_report.py [--all] [file.hip ...]   (default: every source build.py compiles)
Prints the kernels that use scratch memory (or all with --all) and exits 1 if any kernel of the default library does."""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import importlib.util
spec = importlib.util.spec_from_file_location('hftt_build', os.path.join(ROOT, 'nylon-amt_amd', 'build.py'))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
show_all = '--all' in sys.argv
files = [a for a in sys.argv[1:] if not a.startswith('--')] or [os.path.join(B.CSRC, s) for s in B.SOURCES if s.endswith('.hip')]


def analyse(src):
    out = subprocess.run([B._hipcc()] + B.FLAGS + ['-x', 'hip', '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
    if out.returncode != 0:
        return src, None, out.stderr[-2000:]
    rows, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1); rows[cur] = {}
            continue
        m = re.search(r'remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)', line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    names = subprocess.run(['c++filt'], input='\n'.join(rows), capture_output=True, text=True).stdout.splitlines()
    return src, [(re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', ''), v) for n, v in zip(names, rows.values())], None


bad = 0
with ThreadPoolExecutor(max_workers=8) as ex:
    for src, rows, err in ex.map(analyse, files):
        print('==', os.path.relpath(src, ROOT))
        if rows is None:
            print(err); bad += 1
            continue
        n_scr = 0
        for name, v in rows:
            scr = v.get('ScratchSize [bytes/lane]', 0)
            if scr or show_all:
                print('  %-64s vgpr %3s agpr %3s  spilled vgpr %3s sgpr %3s  scratch %4s B/lane  occupancy %s' % (
                    name[:64], v.get('VGPRs'), v.get('AGPRs'), v.get('VGPRs Spill'), v.get('SGPRs Spill'), scr, v.get('Occupancy [waves/SIMD]')))
            n_scr += 1 if scr else 0
        print('  %d kernels, %d with scratch' % (len(rows), n_scr))
        bad += n_scr
print('TOTAL kernels with scratch: %d' % bad)
sys.exit(1 if bad else 0)
