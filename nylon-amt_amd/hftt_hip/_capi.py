"""ctypes binding of libhftt_hip.so (the C ABI declared in include/hftt_hip.h).

The library is the product path: if it cannot be loaded this module raises at import of the symbols
(``lib()``) -- there is NO CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = 'libhftt_hip_g' if os.environ.get('HFTT_BUILD_GRAD_HI') == '1' else 'libhftt_hip'
LIB_PATH = os.environ.get('HFTT_LIB_PATH') or os.path.join(_HERE, '..', 'lib', _LIB_NAME + '.so')      # HFTT_LIB_PATH: dev builds (tools/ablate_strip.sh)

c_f32p = C.c_void_p   # device pointers travel as plain integers
c_u16p = C.c_void_p


class PrepEntry(C.Structure):
    _fields_ = [('src_off', C.c_int64), ('dst_off', C.c_int64),
                ('rows', C.c_int32), ('cols', C.c_int32), ('src_ld', C.c_int32), ('dst_ld', C.c_int32),
                ('kind', C.c_int32), ('pad', C.c_int32)]


class GemmNtDesc(C.Structure):
    _fields_ = [('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('npass', C.c_int32),
                ('A', c_f32p), ('lda', C.c_int64),
                ('W', C.c_void_p), ('io_flags', C.c_uint32), ('debug', C.c_uint32),
                ('bias', c_f32p),
                ('C', c_f32p), ('ldc', C.c_int64),
                ('act', C.c_int32), ('out_scale', C.c_float),
                ('add_table', c_f32p), ('add_mod', C.c_int32),
                ('gate', c_f32p), ('ldg', C.c_int64), ('gate_scale', C.c_float),
                ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64),
                ('residual', c_f32p), ('ldr', C.c_int64), ('res_mod', C.c_int32),
                ('ln_gamma', c_f32p), ('ln_beta', c_f32p), ('pre_ln_out', c_f32p), ('ln_mean', c_f32p), ('ln_rstd', c_f32p),
                ('W_lo', C.c_void_p)]


class GemmTnDesc(C.Structure):
    _fields_ = [('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('npass', C.c_int32),
                ('dY', c_f32p), ('lddy', C.c_int64),
                ('X', c_f32p), ('ldx', C.c_int64),
                ('out_scale', C.c_float), ('beta', C.c_float),
                ('n_seg', C.c_int32),
                ('seg_row0', C.c_int32 * 8), ('seg_rows', C.c_int32 * 8),
                ('seg_dw', c_f32p * 8), ('seg_db', c_f32p * 8),
                ('K_out', C.c_int32), ('io_flags', C.c_uint32),
                ('ws', C.c_void_p), ('ws_bytes', C.c_int64),
                ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64)]


class AttnDesc(C.Structure):
    _fields_ = [('n_seq', C.c_int32), ('n_heads', C.c_int32), ('Lq', C.c_int32), ('Lk', C.c_int32), ('dh', C.c_int32), ('npass', C.c_int32),
                ('q', c_f32p), ('q_seq_stride', C.c_int64), ('ldq', C.c_int64),
                ('k', c_f32p), ('k_seq_stride', C.c_int64), ('ldk', C.c_int64),
                ('v', c_f32p), ('v_seq_stride', C.c_int64), ('ldv', C.c_int64),
                ('out', c_f32p), ('o_seq_stride', C.c_int64), ('ldo', C.c_int64),
                ('lse', c_f32p), ('probs', c_f32p),
                ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64),
                ('dout', c_f32p),
                ('dq', c_f32p), ('dq_seq_stride', C.c_int64), ('lddq', C.c_int64),
                ('dk', c_f32p), ('dk_seq_stride', C.c_int64), ('lddk', C.c_int64),
                ('dv', c_f32p), ('dv_seq_stride', C.c_int64), ('lddv', C.c_int64),
                ('io_flags', C.c_uint32), ('pad', C.c_uint32)]


class FoldDesc(C.Structure):
    _fields_ = [('d', C.c_int32), ('C', C.c_int32), ('kw', C.c_int32), ('n_proc', C.c_int32), ('Kp', C.c_int32), ('d_pad', C.c_int32),
                ('wconv', c_f32p), ('bconv', c_f32p), ('wtok', c_f32p), ('btok', c_f32p),
                ('weff_bf', c_u16p), ('weff_f32', c_f32p), ('beff', c_f32p),
                ('dweff', c_f32p), ('dbeff', c_f32p),
                ('g_wconv', c_f32p), ('g_bconv', c_f32p), ('g_wtok', c_f32p), ('g_btok', c_f32p),
                ('weff_hi', c_u16p), ('weff_lo', c_u16p)]


class LnBwdDesc(C.Structure):
    _fields_ = [('M', C.c_int32), ('N', C.c_int32),
                ('dy', c_f32p), ('r', c_f32p), ('mean', c_f32p), ('rstd', c_f32p), ('gamma', c_f32p),
                ('dr', c_f32p), ('dr_drop', c_f32p),
                ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64),
                ('ws', c_f32p), ('drop_bf16', C.c_uint32), ('io_flags', C.c_uint32)]


class StripPackEntry(C.Structure):
    _fields_ = [('src_off', C.c_int64), ('dst_off', C.c_int64),
                ('rows', C.c_int32), ('cols', C.c_int32), ('src_ld', C.c_int32), ('transpose', C.c_int32),
                ('n0', C.c_int32), ('k0', C.c_int32), ('K', C.c_int32), ('order', C.c_int32),
                ('slot_stride', C.c_int32), ('slot_offset', C.c_int32)]


class StripDesc(C.Structure):
    _fields_ = [('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('flags', C.c_uint32),
                ('x', C.c_void_p), ('ldx', C.c_int64),
                ('w', C.c_void_p), ('bias', c_f32p),
                ('C', C.c_void_p), ('ldc', C.c_int64),
                ('out_scale', C.c_float), ('gate_scale', C.c_float),
                ('gate', C.c_void_p), ('ldg', C.c_int64),
                ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64),
                ('residual', C.c_void_p), ('ldr', C.c_int64), ('res_mod', C.c_int32), ('pad', C.c_int32),
                ('ln_gamma', c_f32p), ('ln_beta', c_f32p), ('pre_ln_out', C.c_void_p), ('ln_mean', c_f32p), ('ln_rstd', c_f32p)]


class FfnDesc(C.Structure):
    _fields_ = [('M', C.c_int32), ('d', C.c_int32), ('p', C.c_int32), ('flags', C.c_uint32),
                ('mode', C.c_int32), ('pad', C.c_int32),
                ('x', C.c_void_p), ('ldx', C.c_int64),
                ('w', C.c_void_p), ('b1', c_f32p), ('b2', c_f32p),
                ('h_out', C.c_void_p), ('ldh', C.c_int64),
                ('gate', C.c_void_p), ('ldg', C.c_int64),
                ('gate_scale', C.c_float),
                ('drop_p', C.c_float), ('site_h', C.c_uint32), ('site_o', C.c_uint32), ('drop_seed', C.c_uint64),
                ('residual', C.c_void_p), ('ldr', C.c_int64),
                ('ln_gamma', c_f32p), ('ln_beta', c_f32p), ('pre_ln_out', C.c_void_p), ('ln_mean', c_f32p), ('ln_rstd', c_f32p),
                ('y', C.c_void_p), ('ldy', C.c_int64)]


SL_X_BF16, SL_C_BF16, SL_RES_BF16, SL_RELU = 1, 2, 4, 8
SL_PRE_BF16 = 128                           # split modes, LayerNorm forms: pre_ln_out stored as bf16
SL_X3_GRAD_HI = 256                         # with SL_X3_BF16: the gradient strip enters as its bf16 rounding (two MFMA passes)
SL_H_BF16 = 64                              # fused block in a split mode: h_out / gate stored as bf16
SL_X3_F16, SL_X3_BF16 = 16, 32               # split-operand forms of the strip kernels (fp32 tensors; fp16 / bf16 hi + lo halves)
SL_X_DROP = 2048                            # backward strip linear: x (the gradient of a dropout output) is masked on load
SL_C_F16PAIR = 512                          # SL_X3_F16 projection: C written as f16-pair planes (the attention kernels' operand form)
ATTN_Q_F16PAIR, ATTN_KV_F16PAIR = 32, 64    # npass 2, dh 64: q / k, v are f16-pair planes
TE_X_BF16, TE_Y_BF16, TE_M_BF16 = 1, 2, 4          # hftt_time_embed_fwd / _bwd io_flags


class LossDesc(C.Structure):
    _fields_ = [('n', C.c_int64), ('V', C.c_int32), ('pad', C.c_int32),
                ('prob', c_f32p * 6), ('vel', c_f32p * 2),
                ('label_onset', c_f32p), ('label_offset', c_f32p), ('label_mpe', c_f32p), ('label_velocity', C.c_void_p),
                ('weight_A', C.c_float), ('weight_B', C.c_float), ('grad_scale', C.c_float), ('pad2', C.c_float),
                ('d_prob', c_f32p * 6), ('d_vel', c_f32p * 2),
                ('loss_out', c_f32p), ('ws', c_f32p)]


class LogmelDesc(C.Structure):
    _fields_ = [('wave', c_f32p), ('n_samples', C.c_int64),
                ('n_fft', C.c_int32), ('hop', C.c_int32), ('n_mels', C.c_int32), ('n_frames', C.c_int32),
                ('window', c_f32p), ('twiddle', c_f32p),
                ('fb_start', C.c_void_p), ('fb_len', C.c_void_p), ('fb_off', C.c_void_p), ('fb_w', c_f32p),
                ('log_offset', C.c_float), ('pad', C.c_int32),
                ('feat', c_f32p)]


class ResampleDesc(C.Structure):
    _fields_ = [('wave', c_f32p), ('n_in', C.c_int64), ('kernel', c_f32p),
                ('up', C.c_int32), ('down', C.c_int32), ('width', C.c_int32), ('taps', C.c_int32),
                ('out', c_f32p), ('n_out', C.c_int64)]


# name -> (restype, argtypes); every symbol include/hftt_hip.h declares
SIGNATURES = {
    'hftt_abi_version': (C.c_int, []),
    'hftt_last_error': (C.c_char_p, []),
    'hftt_build_options': (C.c_int, []),
    'hftt_device_cus': (C.c_int, []),
    'hftt_prep_weights': (C.c_int, [c_f32p, c_u16p, c_u16p, c_f32p, C.c_void_p, C.c_int, C.c_void_p]),
    'hftt_prep_weights_x3': (C.c_int, [c_f32p, c_u16p, c_u16p, c_f32p, C.c_void_p, C.c_int, C.c_void_p]),
    'hftt_gemm_nt': (C.c_int, [C.POINTER(GemmNtDesc), C.c_void_p]),
    'hftt_strip_pack': (C.c_int, [c_f32p, c_u16p, C.c_void_p, C.c_int, C.c_void_p]),
    'hftt_x3_strip_pack': (C.c_int, [c_f32p, c_u16p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'hftt_strip_linear': (C.c_int, [C.POINTER(StripDesc), C.c_void_p]),
    'hftt_ffn_res_ln_fwd': (C.c_int, [C.POINTER(FfnDesc), C.c_void_p]),
    'hftt_ffn_bwd_dx': (C.c_int, [C.POINTER(FfnDesc), C.c_void_p]),
    'hftt_attn_out_ffn_fwd': (C.c_int, [C.POINTER(StripDesc), C.POINTER(FfnDesc), C.c_void_p]),
    'hftt_gemm_tn_ws_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'hftt_gemm_tn': (C.c_int, [C.POINTER(GemmTnDesc), C.c_void_p]),
    'hftt_attn_fwd': (C.c_int, [C.POINTER(AttnDesc), C.c_void_p]),
    'hftt_attn_bwd': (C.c_int, [C.POINTER(AttnDesc), C.c_void_p]),
    'hftt_x3_to_planes': (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    'hftt_embed_fold_fwd': (C.c_int, [C.POINTER(FoldDesc), C.c_void_p]),
    'hftt_embed_fold_bwd': (C.c_int, [C.POINTER(FoldDesc), C.c_void_p]),
    'hftt_im2win': (C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'hftt_ln_bwd_wgs': (C.c_int32, [C.c_int32]),
    'hftt_ln_bwd': (C.c_int, [C.POINTER(LnBwdDesc), C.c_void_p]),
    'hftt_ln_bwd_reduce': (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, C.c_float, C.c_void_p]),
    'hftt_time_embed_fwd': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_float, C.c_float, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]),
    'hftt_time_embed_bwd': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_float, C.c_float, C.c_uint32, C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p]),
    'hftt_dropout_bwd': (C.c_int, [c_f32p, C.c_int64, C.c_float, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]),
    'hftt_colsum': (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, C.c_float, c_f32p, C.c_uint32, C.c_void_p]),
    'hftt_colsum_ws_bytes': (C.c_int64, [C.c_int64, C.c_int64]),
    'hftt_heads_split': (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p,
                                   C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'hftt_heads_split_bwd': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'hftt_loss_ws_bytes': (C.c_int64, [C.c_int64]),
    'hftt_loss': (C.c_int, [C.POINTER(LossDesc), C.c_void_p]),
    'hftt_adam_step': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32,
                                 C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    'hftt_logmel': (C.c_int, [C.POINTER(LogmelDesc), C.c_void_p]),
    'hftt_resample': (C.c_int, [C.POINTER(ResampleDesc), C.c_void_p]),
}

_lib = None
ABI_VERSION = 8


class HfttError(RuntimeError):
    pass


def lib():
    """Load libhftt_hip.so once; raise loudly if it is missing (no fallback path exists)."""
    global _lib
    if _lib is None:
        path = os.path.abspath(LIB_PATH)
        if not os.path.exists(path):
            raise HfttError('libhftt_hip.so not found at %s -- build it with `python nylon-amt_amd/build.py` '
                            '(the HIP library is the only compute path; there is no CPU fallback)' % path)
        # torch first: the library's streams and device pointers come from torch, so both must sit on ONE HIP runtime.  Loaded before torch, the
        # library pulls in /opt/rocm's libamdhip64 and torch then brings its own copy -- every launch fails with "no ROCm-capable device".
        import torch   # noqa: F401
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is missing: also loud
            fn.restype = res
            fn.argtypes = args
        if handle.hftt_abi_version() != ABI_VERSION:
            raise HfttError('libhftt_hip.so ABI version mismatch')
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().hftt_last_error()
        raise HfttError('%s failed (rc=%d): %s' % (what or 'hftt call', rc, msg.decode() if msg else '?'))
