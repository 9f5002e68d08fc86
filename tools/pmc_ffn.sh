#!/bin/bash
# GPU box: clock, matrix-pipe busy fraction and wave-cycle split of the bf16 fused feed-forward block at S_e, one and two workgroups per CU
# (tools/bench_strip.py ffn under one rocprofv3 --pmc pass each; the program sits directly behind `--`).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_ffn
rm -rf $OUT; mkdir -p $OUT
for wpc in 1 2; do
  export HFTT_MLP2_WPC=$wpc
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/w$wpc -o p --output-format csv -- python3 $R/tools/bench_strip.py ffn > $OUT/w$wpc.log 2>&1 || { tail -5 $OUT/w$wpc.log; exit 1; }
done
python3 - $OUT <<'PY'
import collections, csv, glob, re, sys
out = sys.argv[1]
for wpc in (1, 2):
    cc = glob.glob('%s/w%d/**/*counter_collection.csv' % (out, wpc), recursive=True)
    kt = glob.glob('%s/w%d/**/*kernel_trace.csv' % (out, wpc), recursive=True)
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp']), r['Kernel_Name'])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        if 'strip_mlp2' not in r['Kernel_Name']:
            continue
        acc[r['Dispatch_Id']][r['Counter_Name']].append(float(r['Counter_Value']))
    rows = []
    for did, c in acc.items():
        c = {k: sum(v) for k, v in c.items()}
        ns = dur[did][0]
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        w = c['SQ_WAVE_CYCLES'] or 1
        rows.append((ns / 1e3, cyc / ns, c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), c['SQ_WAIT_ANY'] / w, c['SQ_WAIT_INST_ANY'] / w, c['SQ_ACTIVE_INST_ANY'] / w, c['SQ_ACTIVE_INST_VALU'] / w))
    # launches alternate: training form (13 launches), inference form (13): report the median of each half by duration class
    rows.sort()
    half = len(rows) // 2
    for name, part in (('inference form', rows[:half]), ('training form ', rows[half:])):
        m = part[len(part) // 2]
        print('WPC=%d %s  %7.1f us  clock %.2f GHz  mfma_busy %.3f  | wave-cycles: waiting %.2f  issue-stalled %.2f  issuing %.2f (VALU %.2f)' % ((wpc, name) + m))
PY
