#!/usr/bin/env python3
"""Summarise tools/pmc_strip.sh: per bench_strip case (13 consecutive dispatches of one kernel symbol) the mean of each counter."""
import collections, csv, glob, re, sys
out = sys.argv[1]
rows = collections.OrderedDict()       # (pass, dispatch id) -> (kernel, {counter: value})
for sub in ('fetch', 'write', 'sq'):
    files = glob.glob('%s/%s/**/*counter_collection.csv' % (out, sub), recursive=True)
    if not files:
        print('no counter file for', sub); continue
    seq = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        k = int(r['Dispatch_Id'])
        seq.setdefault(k, [r['Kernel_Name'], {}])[1][r['Counter_Name']] = float(r['Counter_Value'])
    # chunk consecutive dispatches of the same strip kernel
    chunks, cur = [], None
    for k, (name, vals) in seq.items():
        m = re.search(r'::(strip_[a-z0-9_]+<[^>]*>)', name)
        key = m.group(1) if m else None
        if key is None or 'pack' in name:
            cur = None
            continue
        if cur is None or cur[0] != key or len(cur[1]) >= 13:
            cur = [key, []]
            chunks.append(cur)
        cur[1].append(vals)
    rows[sub] = chunks
n = max(len(v) for v in rows.values())
for i in range(n):
    line = []
    name = None
    for sub, chunks in rows.items():
        if i >= len(chunks):
            continue
        name = chunks[i][0]
        ctrs = collections.defaultdict(list)
        for vals in chunks[i][1][3:]:          # skip the 3 warm-up launches
            for c, v in vals.items():
                ctrs[c].append(v)
        for c, v in ctrs.items():
            mean = sum(v) / len(v)
            if c == 'FETCH_SIZE':
                line.append('fetch(x2)=%.0fMB' % (2 * mean * 1024 / 1e6))
            elif c == 'WRITE_SIZE':
                line.append('write=%.0fMB' % (mean * 1024 / 1e6))
            else:
                line.append('%s=%.3g' % (c.replace('SQ_', ''), mean))
    print('%-2d %-42s %s' % (i, name, '  '.join(line)))
