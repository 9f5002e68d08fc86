#!/bin/bash
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_model_gpu.py tests/test_paper_bf16_gpu.py tests/test_kernel_names_gpu.py tests/test_train_replay_gpu.py -k "golden_fixture or paper or dropout_on_outputs or mini_full or other_configurations or kernel_names or buckets or data_parallel" -q -s -p no:cacheprovider > $O/r06_ckvb.log 2>&1; echo "rc=$?" >> $O/r06_ckvb.log
grep "passed\|failed\|rc=\|FAILED\|paper_b1 x3" $O/r06_ckvb.log | cut -c1-400
grep -q "rc=0" $O/r06_ckvb.log || exit 0
for v in 1 0 1 0 1 0; do
  HFTT_X3_MERGE_CKV_BWD=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-extras --no-pmc --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge_bwd=$v: %.1f clips/s  step ms min/median/max %.2f %.2f %.2f' % (j['value'], j['step_ms_min'], j['step_ms_median'], j['step_ms_max']))"
done > $O/r06_ab_merge_ckv_bwd.txt 2>&1
cat $O/r06_ab_merge_ckv_bwd.txt
