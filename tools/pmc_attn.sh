#!/bin/bash
# GPU box: SQ / LDS counters of the x3 attention launches of tools/bench_x3_attn.py (two passes of eight counters; the program sits directly
# behind `--`).  Prints the per-kernel mean of every counter.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
export CROSS=${CROSS:-0}
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq -o p --output-format csv -- python3 $R/tools/bench_x3_attn.py > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT -d $OUT/lds -o p --output-format csv -- python3 $R/tools/bench_x3_attn.py > $OUT/lds.log 2>&1 || { tail -5 $OUT/lds.log; exit 1; }
python3 - $OUT <<'PY'
import collections, csv, glob, re, sys
out = sys.argv[1]
for sub in ('sq', 'lds'):
    files = glob.glob('%s/%s/**/*counter_collection.csv' % (out, sub), recursive=True)
    if not files:
        print('no counter file for', sub); continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        m = re.search(r'::(x3p?_attn_[a-z]+_kernel<[^>]*>)', r['Kernel_Name'])
        if not m: continue
        acc.setdefault(m.group(1), collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in acc.items():
        print('%-4s %-44s %s' % (sub, k, '  '.join('%s=%.4g' % (n.replace('SQ_', ''), sum(v) / len(v)) for n, v in c.items())))
PY
