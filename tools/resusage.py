#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print per-kernel register/LDS/spill usage (dev tool)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-c', src, '-o', '/tmp/_res.o',
                      '-Rpass-analysis=kernel-resource-usage'] + sys.argv[2:], capture_output=True, text=True)
txt = out.stderr
if out.returncode != 0:
    print(txt[-4000:]); sys.exit(1)
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r'\(anonymous namespace\)::', '', cur).split('(')[0]
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    print(f"{k:60s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} spill={v.get('VGPRs Spill')} scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')} sgpr_spill={v.get('SGPRs Spill')}")
for l in txt.splitlines():
    if 'warning' in l or 'error' in l: print(l)
