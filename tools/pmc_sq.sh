#!/bin/bash
# GPU box: SQ / LDS counters per kernel of a few training steps (bench.py without its extra legs), two passes of eight counters; the program
# sits directly behind `--`.  Prints, per kernel symbol, launches and the SUM of every counter, sorted by SQ_WAVE_CYCLES.
#   bash tools/pmc_sq.sh [paper|tiny]
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-paper}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
B="python3 $R/bench.py --steps 2 --warmup 1 --config $CFG --no-cpu-baseline --no-profile --no-extras --no-pmc"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq -o p --output-format csv -- $B > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU -d $OUT/lds -o p --output-format csv -- $B > $OUT/lds.log 2>&1 || { tail -5 $OUT/lds.log; exit 1; }
python3 - $OUT <<'PY'
import collections, csv, glob, re, sys
out = sys.argv[1]
acc = collections.OrderedDict()
for sub in ('sq', 'lds'):
    files = glob.glob('%s/%s/**/*counter_collection.csv' % (out, sub), recursive=True)
    if not files:
        print('no counter file for', sub); continue
    seen = set()
    for r in csv.DictReader(open(files[0])):
        name = re.sub(r'^void \(anonymous namespace\)::|^\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'\(.*$', '', name)
        d = acc.setdefault(name, collections.defaultdict(float))
        d[r['Counter_Name']] += float(r['Counter_Value'])
        if sub == 'sq' and (r['Dispatch_Id'], name) not in seen:
            seen.add((r['Dispatch_Id'], name)); d['launches'] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))
tot = sum(v.get('SQ_WAVE_CYCLES', 0) for _, v in rows)
print('%-58s %5s %6s | %5s %5s %5s | %5s %5s | %6s %6s' % ('kernel', 'n', 'wave%', 'wait', 'stall', 'activ', 'valu', 'lds', 'ldsact', 'confl'))
for k, v in rows[:40]:
    w = v.get('SQ_WAVE_CYCLES', 0) or 1
    print('%-58s %5d %6.2f | %5.2f %5.2f %5.2f | %5.2f %5.2f | %6.3g %6.2f   valu/mfma %.1f' % (
        k[:58], v['launches'], 100 * w / tot, v['SQ_WAIT_ANY'] / w, v['SQ_WAIT_INST_ANY'] / w, v['SQ_ACTIVE_INST_ANY'] / w,
        v['SQ_ACTIVE_INST_VALU'] / w, v['SQ_ACTIVE_INST_LDS'] / w, v['SQ_LDS_IDX_ACTIVE'],
        v['SQ_LDS_BANK_CONFLICT'] / (v['SQ_LDS_IDX_ACTIVE'] or 1), v['SQ_INSTS_VALU'] / (v['SQ_INSTS_MFMA'] or 1)))
PY
