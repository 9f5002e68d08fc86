"""The kernel symbols the engine writes into its launch plans (hftt_hip/engine.py plan meta: `kernel`) are the keys under which bench.py looks up
hardware counters -- HBM traffic and MFMA-busy per kernel of THIS run, and the committed rocprofv3 summaries under profiles/.  A symbol that
drifts from what rocprofv3 prints makes the line quote another build's counters without anyone noticing (the bf16 fused
FFN had gained two template arguments and the line fell through to a round-3 profile).  Here every plan-meta name of both precision modes,
training and inference plans, must occur verbatim in a kernel trace of the same process."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

import pytest

import util

pytestmark = pytest.mark.gpu


def _trace(tmp_path, extra, tag):
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        pytest.skip('rocprofv3 is not on this box')
    sys.path.insert(0, util.ROOT)
    import bench
    out = str(tmp_path / tag)
    plan = str(tmp_path / (tag + '_plan.json'))
    env = dict(os.environ, TMPDIR='/tmp', HFTT_BENCH_PLAN_DUMP=plan)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    # (the profiled program itself follows `--`: no shell, env or launcher in between)
    cmd = [exe, '--kernel-trace', '-d', out, '-o', 't', '--output-format', 'csv', '--', 'python3', os.path.join(util.ROOT, 'bench.py'),
           '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-profile', '--no-extras', '--no-pmc'] + extra
    r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    files = glob.glob(os.path.join(out, '**', 't_kernel_trace.csv'), recursive=True)
    assert files, os.listdir(out)
    traced = {bench._kernel_key(row['Kernel_Name']) for row in csv.DictReader(open(files[0]))}
    return traced, json.load(open(plan))


@pytest.mark.parametrize('config', ['paper', 'tiny'])
@pytest.mark.parametrize('precision', ['x3', 'bf16'])
def test_training_plan_kernel_names_are_the_traced_symbols(dev, tmp_path, config, precision):
    traced, plans = _trace(tmp_path, ['--config', config, '--precision', precision], 'train')
    names = {e[0] for e in plans['train:%s' % precision]}
    assert len(names) >= 8
    missing = sorted(names - traced)
    assert not missing, (missing, sorted(t for t in traced if not t.startswith('(other)')))
    assert 'adam_kernel' in traced and 'im2win_kernel' in traced      # the step separators of bench.py::measure_pmc


def test_inference_plan_kernel_names_are_the_traced_symbols(dev, tmp_path):
    """`bench.py --inference-only`: three eval forwards in the x3 mode, then three in the bf16 mode (the command measure_pmc wraps for the
    counters of `roofline_ffn` and `bf16_mode.roofline_ffn`)"""
    traced, plans = _trace(tmp_path, ['--inference-only'], 'inf')
    for mode in ('x3', 'bf16'):
        names = {e[0] for e in plans['inference:%s' % mode]}
        missing = sorted(names - traced)
        assert not missing, (mode, missing)
        ffn = [e for e in plans['inference:%s' % mode] if '_mlp' in e[0]]
        assert len(ffn) == 9 and ffn[0][1][0] == 8 * 128 * 256          # three encoder launches at S_e, six at S_n
    # (x3: fc_o + LayerNorm + FFN as one launch, the default since round 6 -- or the FFN block alone with HFTT_X3_FUSE_OFFN=0)
    assert any(n.startswith('strip_mlp2_kernel<0, 16, false>') for n in traced) and any(n.startswith(('x3_oln_mlp_kernel<', 'x3_mlp_kernel<0, 16')) for n in traced)
