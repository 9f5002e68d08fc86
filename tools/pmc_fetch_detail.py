#!/usr/bin/env python3
"""Reduce tools/pmc_fetch_detail.sh: per kernel symbol, the bytes the L2 fetched from the fabric by request size (exact: 32 / 64 / 128-byte
requests), next to what `FETCH_SIZE x 2` would have said, the L2 hit rate and the bytes written; writes profiles/<tag>_pmc_fetch_detail.json."""
import collections, csv, glob, json, os, re, sys
tag = sys.argv[1]
G = 'gpurun_out/' + tag


def key_of(n):
    m = re.search(r'::([a-z_0-9]+(?:<[^>]*>)?)\(', n)
    return m.group(1) if m else n


def load(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', 'b_counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'namespace)::' in n and 'at::' not in n:
                out[key_of(n)][r['Counter_Name']].append(float(r['Counter_Value']))
    return out


rd, l2 = load(G + '_rdreq'), load(G + '_l2')
rows = {}
for k in sorted(rd):
    c = rd[k]
    n = len(c['TCC_EA0_RDREQ_sum'])
    tot, r32, r64, r128 = (sum(c[x]) / n for x in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_EA0_RDREQ_64B_sum', 'TCC_EA0_RDREQ_128B_sum'))
    other = tot - r32 - r64 - r128
    exact = 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(other, 0.0)
    e = {'launches': n, 'fetch_MB_exact': round(exact / 1e6, 2), 'fetch_MB_as_FETCH_SIZE_x2': round(2 * 64 * tot / 1e6, 2),
         'requests': {'32B': round(r32), '64B': round(r64), '128B': round(r128), 'unclassified': round(other)}}
    if k in l2:
        h, m = sum(l2[k]['TCC_HIT_sum']) / len(l2[k]['TCC_HIT_sum']), sum(l2[k]['TCC_MISS_sum']) / len(l2[k]['TCC_MISS_sum'])
        e['l2_hit_rate'] = round(h / max(h + m, 1.0), 4)
        w, w64 = sum(l2[k]['TCC_EA0_WRREQ_sum']) / len(l2[k]['TCC_EA0_WRREQ_sum']), sum(l2[k]['TCC_EA0_WRREQ_64B_sum']) / len(l2[k]['TCC_EA0_WRREQ_64B_sum'])
        e['write_MB'] = round((64 * w64 + 32 * (w - w64)) / 1e6, 2)
    rows[k] = e
out = {'note': 'rocprofv3 --pmc TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum (pass 1) and TCC_HIT / TCC_MISS / TCC_EA0_WRREQ{,_64B}_sum (pass 2) around '
               '`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras`; per-launch averages per kernel symbol. '
               'fetch_MB_exact = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B; fetch_MB_as_FETCH_SIZE_x2 = what the FETCH_SIZE x 2 rule of '
               'the traffic summaries reports for the same launch (every request at 64 B, doubled)', 'kernels': rows}
json.dump(out, open('profiles/%s_pmc_fetch_detail.json' % tag, 'w'), indent=1)
for k, e in sorted(rows.items(), key=lambda kv: -kv[1]['launches'] * kv[1]['fetch_MB_exact'])[:16]:
    print('%-46s n=%3d exact %8.1f MB  (x2 rule %8.1f)  32B/64B/128B %s  L2 hit %s  write %s' % (k, e['launches'], e['fetch_MB_exact'], e['fetch_MB_as_FETCH_SIZE_x2'],
          '/'.join(str(e['requests'][s]) for s in ('32B', '64B', '128B')), e.get('l2_hit_rate'), e.get('write_MB')))
