"""Compile-time resource checks (hipcc cross-compiles without a GPU): the register allocation the launch plans rely on.

Round 5 found the <= 128-key attention backward forms at HALF their intended occupancy for three rounds: nothing bounded their registers, hipcc
spread them over 280-300 (accumulation registers as spill space), and one workgroup per CU ran where the LDS budget was laid out for two.  The
kernels were correct, so no parity test could see it; this one reads the compiler's own resource remarks."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'nylon-amt_amd', 'csrc')
HIPCC = '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else shutil.which('hipcc')


def _resources(src):
    out = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result', '-x', 'hip', '-c', os.path.join(CSRC, src),
                          '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
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
    return {re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', n): v for n, v in zip(names, rows.values())}


def _get(v, key):
    for k, x in v.items():
        if k.startswith(key):
            return x
    raise KeyError(key)


@pytest.mark.skipif(HIPCC is None, reason='hipcc not found')
def test_attention_backward_on_planes_registers_and_occupancy():
    res = _resources('x3_attn_pl.hip')
    seen = 0
    for name, v in res.items():
        m = re.match(r'x3_attn_bwd_kernel<(\d+), 64, true, (\d)>', name)
        if not m:
            continue
        seen += 1
        kt = int(m.group(1))
        assert _get(v, 'ScratchSize') == 0, (name, v)
        if kt in (3, 4):        # 96 / 128 keys: 63 / 76 KB of LDS = two workgroups per CU, i.e. two waves per SIMD: at most 256 registers in all
            assert _get(v, 'VGPRs') + _get(v, 'AGPRs') <= 256 and _get(v, 'Occupancy') >= 2, (name, v)
        if kt == 8:             # eight waves = two per SIMD
            assert _get(v, 'VGPRs') + _get(v, 'AGPRs') <= 256, (name, v)
    assert seen == 15           # KT 1, 2, 3, 4, 8 x three dropout forms
    # the forward on planes: its persistent grid is sized by LDS alone (launch_pf: three workgroups per CU at <= 96 keys, two at 128), so the
    # registers must allow that many waves per SIMD without scratch
    fwd = 0
    for name, v in res.items():
        m = re.match(r'x3p_attn_fwd_kernel<(\d+), (\d+), (true|false), (\d)>', name)
        if not m:
            continue
        fwd += 1
        kt, nw = int(m.group(1)), int(m.group(2))
        assert _get(v, 'ScratchSize') == 0, (name, v)
        need = 3 if kt <= 3 else (2 if kt <= 4 or nw == 8 else 1)
        assert _get(v, 'Occupancy') >= need, (name, v)
    assert fwd >= 24
