"""The C-ABI library builds for gfx950, loads on a CPU-only box and exports every symbol include/hftt_hip.h declares;
ctypes struct layouts match the C compiler's.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import util

HDR = os.path.join(util.ROOT, 'include', 'hftt_hip.h')


@pytest.fixture(scope='module')
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location('hftt_build', os.path.join(util.ROOT, 'nylon-amt_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)            # hipcc cross-compiles gfx950 without a GPU
    from hftt_hip import _capi
    return _capi.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from hftt_hip import _capi
    src = open(HDR).read()
    declared = set(re.findall(r'\b(hftt_[a-z0-9_]+)\s*\(', src))
    assert len(declared) >= 25
    assert declared == set(_capi.SIGNATURES.keys()), declared ^ set(_capi.SIGNATURES.keys())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hftt_abi_version() == _capi.ABI_VERSION == 8
    assert lib.hftt_last_error() is not None
    # pure host helpers can be called without a GPU
    assert lib.hftt_gemm_tn_ws_bytes(1000, 256, 256) > 0
    assert lib.hftt_ln_bwd_wgs(100000) == 1024
    assert lib.hftt_loss_ws_bytes(1000) > 0 and lib.hftt_colsum_ws_bytes(10, 100) == 16 * 100 * 4


def test_argument_validation_without_gpu(lib):
    """Descriptor checks run before any launch, so error behaviour is testable on the CPU."""
    from hftt_hip import _capi
    d = _capi.GemmNtDesc()
    assert lib.hftt_gemm_nt(C.byref(d), None) != 0
    assert b'bad shape' in lib.hftt_last_error()
    d.M, d.N, d.K, d.npass = 8, 8, 7, 1
    assert lib.hftt_gemm_nt(C.byref(d), None) != 0
    assert b'multiple of 32' in lib.hftt_last_error()
    a = _capi.AttnDesc()
    a.n_seq, a.n_heads, a.Lq, a.Lk, a.dh, a.npass = 1, 1, 300, 10, 64, 1
    assert lib.hftt_attn_fwd(C.byref(a), None) != 0
    assert b'1..256' in lib.hftt_last_error()
    sd = _capi.StripDesc()
    sd.M, sd.N, sd.K = 128, 200, 256
    assert lib.hftt_strip_linear(C.byref(sd), None) != 0
    assert b'multiple of 256' in lib.hftt_last_error()
    fd = _capi.FfnDesc()
    fd.M, fd.d, fd.p, fd.mode = 128, 128, 512, 0
    assert lib.hftt_ffn_res_ln_fwd(C.byref(fd), None) != 0
    assert b'd == 256' in lib.hftt_last_error()
    assert lib.hftt_ffn_bwd_dx(C.byref(fd), None) != 0          # mode 0 descriptor handed to the mode 1 entry point
    # the small-width family (d = 64, ff = 128) keeps the all-bf16 storage requirement of the fused block: an fp32 descriptor of that shape
    # is rejected before any launch, for both entry points (it used to be reinterpreted as bf16)
    for mode, fn in ((0, lib.hftt_ffn_res_ln_fwd), (1, lib.hftt_ffn_bwd_dx)):
        fs = _capi.FfnDesc()
        fs.M, fs.d, fs.p, fs.mode, fs.flags = 128, 64, 128, mode, 0
        fs.x = fs.w = fs.y = 0x1000
        assert fn(C.byref(fs), None) != 0
        assert b'all-bf16' in lib.hftt_last_error(), lib.hftt_last_error()
    # the masked weight-gradient loader indexes its dropout site with 32-bit hash quads: a site beyond that range is refused, not wrapped
    td = _capi.GemmTnDesc()
    td.M, td.N, td.K, td.npass, td.lddy, td.ldx, td.n_seg = 1 << 24, 256, 256, 4, 256, 256, 1
    td.io_flags, td.drop_p, td.K_out, td.out_scale = 8, 0.1, 256, 1.0      # HFTT_TN_DY_DROP
    td.dY = td.X = td.ws = 0x1000
    td.ws_bytes = lib.hftt_gemm_tn_ws_bytes(td.M, td.N, td.K)
    td.seg_dw[0], td.seg_rows[0] = 0x1000, 256
    rc = lib.hftt_gemm_tn(C.byref(td), None)
    assert rc != 0 and b'32-bit quads' in lib.hftt_last_error(), lib.hftt_last_error()
    with pytest.raises(_capi.HfttError):
        _capi.check(1, 'x')


def test_struct_layouts_match_the_c_compiler(lib, tmp_path):
    from hftt_hip import _capi
    structs = {'hftt_prep_entry': _capi.PrepEntry, 'hftt_gemm_nt_desc': _capi.GemmNtDesc, 'hftt_gemm_tn_desc': _capi.GemmTnDesc,
               'hftt_attn_desc': _capi.AttnDesc, 'hftt_fold_desc': _capi.FoldDesc, 'hftt_ln_bwd_desc': _capi.LnBwdDesc,
               'hftt_loss_desc': _capi.LossDesc, 'hftt_logmel_desc': _capi.LogmelDesc, 'hftt_resample_desc': _capi.ResampleDesc,
               'hftt_strip_pack_entry': _capi.StripPackEntry, 'hftt_strip_desc': _capi.StripDesc, 'hftt_ffn_desc': _capi.FfnDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HDR, 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    lines += ['return 0; }']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c99', str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split('\n')
    got = dict(l.split() for l in out if l)
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for f in cls._fields_:
            assert int(got['%s.%s' % (cname, f[0])]) == getattr(cls, f[0]).offset, (cname, f[0])


def test_integration_stub_asserts_the_library_abi_version(lib):
    """INTEGRATION.md's ctypes stub is what a maintainer pastes: its ABI assertion must be the loaded library's version (it said 4 at ABI 6)"""
    txt = open(os.path.join(util.ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'hftt_abi_version\(\)\s*==\s*(\d+)', txt)
    assert m, 'the stub lost its ABI assertion'
    assert int(m.group(1)) == lib.hftt_abi_version()
