#!/usr/bin/env python3
"""Build libhftt_hip.so (gfx950) in-tree with hipcc.  Usage: python nylon-amt_amd/build.py [--force]

Objects are rebuilt only when their source (or a shared header) is newer.  The .so stays in-tree
(git-ignored) so it travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(HERE, 'build')
LIB = os.path.join(LIBDIR, 'libhftt_hip.so')
SOURCES = ['capi.cpp', 'gemm_nt.hip', 'strip_gemm.hip', 'strip_gemm2.hip', 'bs_strip.hip', 'gemm_tn.hip', 'attn_fwd.hip', 'attn_fwd8.hip', 'attn_bwd.hip', 'x3_attn.hip', 'x3_attn_pl.hip', 'x3_strip.hip', 'elementwise.hip', 'logmel.hip']
HEADERS = [os.path.join(CSRC, 'hftt_common.h'), os.path.join(CSRC, 'hftt_host.h'), os.path.join(CSRC, 'strip_internal.h'), os.path.join(CSRC, 'strip_pipe.h'), os.path.join(CSRC, 'x3_common.h'), os.path.join(CSRC, 'x3_internal.h'), os.path.join(CSRC, 'x3_attn_bwd.h'), os.path.join(CSRC, 'x3s_strip.h'),
           os.path.join(HERE, '..', 'include', 'hftt_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result']

# HFTT_BUILD_GRAD_HI=1: also build the opt-in gradient-rounding forms (a gradient operand as its bf16 rounding, two MFMA
# passes; outside the 1e-3 gradient tolerance of the default mode).  They double the x3_linear* instantiations, so the default library leaves
# them out (hftt_build_options() bit 0 says which build is loaded).
if os.environ.get('HFTT_BUILD_GRAD_HI') == '1':
    FLAGS += ['-DHFTT_GRAD_HI_BUILD']
    LIB = os.path.join(LIBDIR, 'libhftt_hip_g.so')


# HFTT_BUILD_TAG=<tag> HFTT_BUILD_EXTRA_FLAGS="...": a side-by-side build for same-box A/B runs (objects *.<tag>.o, library libhftt_hip_<tag>.so,
# selected at run time with HFTT_LIB_PATH); the product library is never touched by it
_TAG = os.environ.get('HFTT_BUILD_TAG', '')
if _TAG:
    FLAGS += os.environ.get('HFTT_BUILD_EXTRA_FLAGS', '').split()
    LIB = os.path.join(LIBDIR, 'libhftt_hip_%s.so' % _TAG)


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.splitext(os.path.basename(src))[0] + (('.g' if 'HFTT_GRAD_HI_BUILD' in ' '.join(FLAGS) else '') + (('.' + _TAG) if _TAG else '') + '.o'))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [_hipcc()] + FLAGS + ['-x', 'hip', '-c', path, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-6000:]))
    return obj


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if verbose:
            print('built', LIB)
    elif verbose:
        print('up to date', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
