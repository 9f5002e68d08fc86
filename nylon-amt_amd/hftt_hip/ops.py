"""Tensor-level wrappers of single C-ABI kernels (used by tests, by ``model.amt`` for the log-mel front end and by
tools).  Each wrapper only builds the POD descriptor from torch tensors and enqueues the kernel on torch's
current stream; results are torch tensors.  No arithmetic happens in Python/torch here.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from ._capi import (AttnDesc, FfnDesc, GemmNtDesc, GemmTnDesc, HfttError, LnBwdDesc, LogmelDesc, LossDesc, ResampleDesc, PrepEntry, StripDesc, StripPackEntry,
                    SL_C_BF16, SL_C_F16PAIR, SL_H_BF16, SL_PRE_BF16, SL_RELU, SL_X3_GRAD_HI, SL_RES_BF16, SL_X_BF16, SL_X3_F16, SL_X3_BF16, SL_X_DROP,
                    ATTN_Q_F16PAIR, ATTN_KV_F16PAIR, check, lib)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != 'cuda':
            raise HfttError('hftt ops need ROCm device tensors (got %s); there is no CPU fallback' % t.device)


BF16 = torch.bfloat16
NT_A_BF16, NT_C_BF16, NT_GATE_BF16, NT_RES_BF16, NT_A_HI = 1, 2, 4, 8, 16
TN_DY_BF16, TN_X_BF16, TN_DY_HI, TN_DY_DROP = 1, 2, 4, 8
ATTN_Q_BF16, ATTN_KV_BF16, ATTN_O_BF16, ATTN_DQ_BF16, ATTN_DKV_BF16 = 1, 2, 4, 8, 16


def _align(x, a):
    return (x + a - 1) // a * a


def prepare_weight(w: torch.Tensor, npass=3, transposed=False, n_pad=64):
    """fp32 [rows, cols] -> the GEMM operand [align(rows, n_pad), cols] (or its transpose): bf16 (int16 tensor) for npass 1,
    fp32 for npass 3; npass 2 / 4 (split fp16 / bf16): an int16 tensor [2, rows_pad, cols] holding the hi plane, then the lo plane."""
    _need_cuda(w)
    w = w.contiguous().float()
    rows, cols = w.shape
    if transposed:
        out_r, out_c = _align(cols, n_pad), _align(rows, 32)
    else:
        out_r, out_c = _align(rows, n_pad), cols
    if npass in (2, 4):
        out = torch.zeros(2, out_r, out_c, dtype=torch.int16, device=w.device)
        ent = (PrepEntry * 1)(PrepEntry(0, 0, rows, cols, cols, out_c, 1 if transposed else 0, npass))
        table = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).to(w.device)
        check(lib().hftt_prep_weights_x3(w.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), 0, table.data_ptr(), 1, _stream(w.device)), 'prep_weights_x3')
        return out
    out = torch.zeros(out_r, out_c, dtype=torch.int16 if npass == 1 else torch.float32, device=w.device)
    ent = (PrepEntry * 1)(PrepEntry(0, 0, rows, cols, cols, out_c, 1 if transposed else 0, 0))
    table = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).to(w.device)
    check(lib().hftt_prep_weights(w.data_ptr(), out.data_ptr() if npass == 1 else 0, out.data_ptr() if npass == 3 else 0, 0,
                                  table.data_ptr(), 1, _stream(w.device)), 'prep_weights')
    return out


def gemm_nt(A, W, bias=None, npass=3, act=0, out_scale=1.0, add_table=None, add_mod=0, gate=None, gate_scale=1.0,
            drop_p=0.0, drop_site=0, drop_seed=0, residual=None, res_mod=0, ln=None, planes=None, debug=0, out_dtype=torch.float32, grad_hi=False):
    """C = epi(A @ W.T + bias); W fp32 [N, K].  ln = (gamma, beta) -> returns (C, pre_ln, mean, rstd).  grad_hi (npass 4): A enters as
    its bf16 rounding (NT_A_HI)."""
    _need_cuda(A, W)
    M, K = A.shape
    N = W.shape[0]
    Wprep = planes if planes is not None else prepare_weight(W, npass)
    Cout = torch.empty(M, N, device=A.device, dtype=out_dtype)
    d = GemmNtDesc()
    d.M, d.N, d.K, d.npass = M, N, K, npass
    d.A, d.lda = A.data_ptr(), A.stride(0)
    d.W = Wprep.data_ptr()
    if npass in (2, 4):
        d.W_lo = Wprep[1].data_ptr()
    d.debug = debug
    d.io_flags = (NT_A_BF16 if A.dtype == BF16 else 0) | (NT_C_BF16 if out_dtype == BF16 else 0) | (NT_GATE_BF16 if (gate is not None and gate.dtype == BF16) else 0) \
        | (NT_RES_BF16 if (residual is not None and residual.dtype == BF16) else 0) | (NT_A_HI if grad_hi else 0)
    d.bias = bias.data_ptr() if bias is not None else 0
    d.C, d.ldc = Cout.data_ptr(), N
    d.act, d.out_scale = act, out_scale
    if add_table is not None:
        d.add_table, d.add_mod = add_table.data_ptr(), add_mod
    if gate is not None:
        d.gate, d.ldg, d.gate_scale = gate.data_ptr(), gate.stride(0), gate_scale
    d.drop_p, d.drop_site, d.drop_seed = drop_p, drop_site, drop_seed
    if residual is not None:
        d.residual, d.ldr, d.res_mod = residual.data_ptr(), residual.stride(0), (res_mod or M)
    extra = ()
    if ln is not None:
        pre = torch.empty(M, N, device=A.device)
        mean = torch.empty(M, device=A.device)
        rstd = torch.empty(M, device=A.device)
        d.ln_gamma, d.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
        d.pre_ln_out, d.ln_mean, d.ln_rstd = pre.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        extra = (pre, mean, rstd)
    check(lib().hftt_gemm_nt(C.byref(d), _stream(A.device)), 'gemm_nt')
    return (Cout,) + extra if extra else Cout


# ---------------------------------------------------------------------------------------------------------------------
# strip kernels (bf16 mode): packed weights, token strips
# ---------------------------------------------------------------------------------------------------------------------
def strip_pack_table(entries, device):
    arr = (StripPackEntry * len(entries))(*[StripPackEntry(*e) for e in entries])
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(entries)


def strip_pack(w: torch.Tensor, transpose=False, order=0, out=None, slot_stride=1, slot_offset=0):
    """fp32 [rows, cols] -> strip-packed bf16 stream of the logical matrix Wl = w (or w.T): int16 tensor of Wl.numel() elements
    (or written into `out` at the given slot stride / offset)."""
    _need_cuda(w)
    w = w.contiguous().float()
    rows, cols = w.shape
    K = rows if transpose else cols
    if out is None:
        out = torch.zeros(w.numel() * slot_stride, dtype=torch.int16, device=w.device)
    table, n = strip_pack_table([(0, 0, rows, cols, cols, 1 if transpose else 0, 0, 0, K, order, slot_stride, slot_offset)], w.device)
    check(lib().hftt_strip_pack(w.data_ptr(), out.data_ptr(), table.data_ptr(), n, _stream(w.device)), 'strip_pack')
    return out


def ffn_pack(w1: torch.Tensor, w2: torch.Tensor, backward=False):
    """Interleaved stream of the fused FFN.  forward: slot 2t = tile t of fc_1.weight [p, d] (tile-major), slot 2t+1 = K-slice t
    of fc_2.weight [d, p].  backward (dX): first matrix = fc_2.weight^T [p, d], second = fc_1.weight^T [d, p]."""
    out = torch.zeros(w1.numel() + w2.numel(), dtype=torch.int16, device=w1.device)
    if not backward:
        strip_pack(w1, False, 1, out, 2, 0)
        strip_pack(w2, False, 0, out, 2, 1)
    else:
        strip_pack(w2, True, 1, out, 2, 0)
        strip_pack(w1, True, 0, out, 2, 1)
    return out


def x3_strip_pack(w: torch.Tensor, elem=2, transpose=False, order=0, out=None, slot_stride=2, slot_offset=0):
    """fp32 [rows, cols] -> split-operand strip stream of the logical matrix Wl = w (or w.T): (hi, lo) fragment pairs of fp16 (elem 2)
    or bf16 (elem 4) halves, 2 * Wl.numel() int16 elements (or written into `out` at the given pair stride / offset)."""
    _need_cuda(w)
    w = w.contiguous().float()
    rows, cols = w.shape
    K = rows if transpose else cols
    if out is None:
        out = torch.zeros(2 * w.numel() * (slot_stride // 2), dtype=torch.int16, device=w.device)
    table, n = strip_pack_table([(0, 0, rows, cols, cols, 1 if transpose else 0, 0, 0, K, order, slot_stride, slot_offset)], w.device)
    check(lib().hftt_x3_strip_pack(w.data_ptr(), out.data_ptr(), table.data_ptr(), n, elem, _stream(w.device)), 'x3_strip_pack')
    return out


def x3_ffn_pack(w1: torch.Tensor, w2: torch.Tensor, backward=False):
    """Interleaved split-operand stream of the fused FFN: per hidden tile two slots of the first matrix, then two of the second."""
    out = torch.zeros(2 * (w1.numel() + w2.numel()), dtype=torch.int16, device=w1.device)
    if not backward:
        x3_strip_pack(w1, 2, False, 1, out, 4, 0)
        x3_strip_pack(w2, 2, False, 0, out, 4, 2)
    else:
        x3_strip_pack(w2, 4, True, 1, out, 4, 0)
        x3_strip_pack(w1, 4, True, 0, out, 4, 2)
    return out


def x3s_pack(w: torch.Tensor, elem=2, transpose=False, nt=None, out=None, pair_offset=0):
    """fp32 [rows, cols] -> COMPACT split-operand pack (order 2) of the logical matrix Wl = w (or w.T) for the small-width strip kernels
    (csrc/x3s_strip.h): the (hi, lo) pair of (k chunk c, output tile t) at pair index pair_offset + c * nt + t, 1024 int16 elements per pair."""
    _need_cuda(w)
    w = w.contiguous().float()
    rows, cols = w.shape
    K, N = (rows, cols) if transpose else (cols, rows)
    nt = nt or N // 32
    if out is None:
        out = torch.zeros((pair_offset + (K // 16) * nt) * 1024, dtype=torch.int16, device=w.device)
    table, n = strip_pack_table([(0, 0, rows, cols, cols, 1 if transpose else 0, 0, 0, K, 2, nt, pair_offset)], w.device)
    check(lib().hftt_x3_strip_pack(w.data_ptr(), out.data_ptr(), table.data_ptr(), n, elem, _stream(w.device)), 'x3_strip_pack')
    return out


def x3s_ffn_pack(w1: torch.Tensor, w2: torch.Tensor, backward=False, bf16=False):
    """the fused block's two matrices at d = 64, p = 128: pairs 0 .. 15 the first GEMM's, 16 .. 31 the second's
    (bf16: bf16 halves for the forward matrices too -- the pack of the bf16 small-width family, csrc/bs_strip.hip, which reads the hi fragments)"""
    out = torch.zeros(32 * 1024, dtype=torch.int16, device=w1.device)
    if not backward:
        x3s_pack(w1, 4 if bf16 else 2, False, 4, out, 0)        # fc_1.weight [p, d]
        x3s_pack(w2, 4 if bf16 else 2, False, 2, out, 16)       # fc_2.weight [d, p]
    else:
        x3s_pack(w2, 4, True, 4, out, 0)         # fc_2.weight^T [p, d]
        x3s_pack(w1, 4, True, 2, out, 16)        # fc_1.weight^T [d, p]
    return out


def strip_linear(x, wpack, N, bias=None, relu=False, out_scale=1.0, gate=None, gate_scale=1.0, drop_p=0.0, drop_site=0, drop_seed=0,
                 residual=None, res_mod=0, ln=None, out_dtype=BF16, save_pre=True, x3=0, pre_bf16=False, grad_hi=False, c_planes=False, x_drop=False):
    """C = epi(x @ Wl.T + bias) with Wl given as its strip pack.  ln = (gamma, beta) -> (C, pre_ln, mean, rstd).
    x3 = 2 / 4: the split-operand form (fp32 tensors, wpack from x3_strip_pack with the same element type).
    x_drop (x3 = 4, N == K == 256): HFTT_SL_X_DROP -- x is the gradient of a dropout output; the mask of (drop_p, drop_site, drop_seed) is applied
    to x while it is loaded (no dropout on the result)."""
    _need_cuda(x, wpack)
    M, K = x.shape
    if x3:
        out_dtype = torch.float32
    Cout = torch.empty(M, N, device=x.device, dtype=out_dtype)
    d = StripDesc()
    d.M, d.N, d.K = M, N, K
    d.flags = (SL_X_BF16 if x.dtype == BF16 else 0) | (SL_C_BF16 if out_dtype == BF16 else 0) | (SL_RELU if relu else 0) \
        | (SL_RES_BF16 if (residual is not None and residual.dtype == BF16) else 0) | (SL_X3_F16 if x3 == 2 else 0) | (SL_X3_BF16 if x3 == 4 else 0) \
        | (SL_PRE_BF16 if (x3 and pre_bf16 and ln is not None) else 0) | (SL_X3_GRAD_HI if (x3 == 4 and grad_hi) else 0) | (SL_C_F16PAIR if c_planes else 0) \
        | (SL_X_DROP if x_drop else 0)
    d.x, d.ldx, d.w = x.data_ptr(), x.stride(0), wpack.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else 0
    d.C, d.ldc, d.out_scale, d.gate_scale = Cout.data_ptr(), N, out_scale, gate_scale
    if gate is not None:
        d.gate, d.ldg = gate.data_ptr(), gate.stride(0)
    d.drop_p, d.drop_site, d.drop_seed = drop_p, drop_site, drop_seed
    if residual is not None:
        d.residual, d.ldr, d.res_mod = residual.data_ptr(), residual.stride(0), res_mod
    extra = ()
    if ln is not None:
        pre = torch.empty(M, N, device=x.device, dtype=BF16 if (x3 and pre_bf16) else out_dtype) if save_pre else None
        mean = torch.empty(M, device=x.device)
        rstd = torch.empty(M, device=x.device)
        d.ln_gamma, d.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
        d.pre_ln_out, d.ln_mean, d.ln_rstd = (pre.data_ptr() if save_pre else 0), mean.data_ptr(), rstd.data_ptr()
        extra = (pre, mean, rstd)
    check(lib().hftt_strip_linear(C.byref(d), _stream(x.device)), 'strip_linear')
    return (Cout,) + extra if extra else Cout


def ffn_res_ln_fwd(x, wpack, p, b1, b2, gamma, beta, drop_p=0.0, site_h=0, site_o=0, seed=0, save_hidden=True, save_pre=True, residual=None, x3=False, hidden_bf16=False, pre_bf16=False):
    """Fused FFN block: y = LN(x + drop(fc_2(drop(relu(fc_1 x))))) -> (y, hidden | None, pre_ln | None, mean, rstd); all bf16
    (x3: all fp32, wpack from x3_ffn_pack; hidden_bf16: the stored hidden as bf16, SL_H_BF16)."""
    _need_cuda(x, wpack)
    M, dm = x.shape
    dt = torch.float32 if x3 else BF16
    y = torch.empty(M, dm, device=x.device, dtype=dt)
    hid = torch.empty(M, p, device=x.device, dtype=BF16 if hidden_bf16 else dt) if save_hidden else None
    pre = torch.empty(M, dm, device=x.device, dtype=BF16 if (x3 and pre_bf16) else dt) if save_pre else None
    mean = torch.empty(M, device=x.device); rstd = torch.empty(M, device=x.device)
    d = FfnDesc()
    d.M, d.d, d.p, d.flags, d.mode = M, dm, p, ((SL_X3_F16 | (SL_H_BF16 if hidden_bf16 else 0) | (SL_PRE_BF16 if pre_bf16 else 0)) if x3 else (SL_X_BF16 | SL_C_BF16 | SL_RES_BF16)), 0
    d.x, d.ldx, d.w = x.data_ptr(), x.stride(0), wpack.data_ptr()
    d.b1, d.b2 = b1.data_ptr(), b2.data_ptr()
    if save_hidden:
        d.h_out, d.ldh = hid.data_ptr(), p
    d.drop_p, d.site_h, d.site_o, d.drop_seed = drop_p, site_h, site_o, seed
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
    d.ln_gamma, d.ln_beta = gamma.data_ptr(), beta.data_ptr()
    d.pre_ln_out = pre.data_ptr() if save_pre else 0
    d.ln_mean, d.ln_rstd = mean.data_ptr(), rstd.data_ptr()
    d.y, d.ldy = y.data_ptr(), dm
    check(lib().hftt_ffn_res_ln_fwd(C.byref(d), _stream(x.device)), 'ffn_res_ln_fwd')
    return y, hid, pre, mean, rstd


def x3_attn_out_ffn_pack(wo: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor):
    """The weight stream of hftt_attn_out_ffn_fwd: fc_o's 16 slots (order 0), then the fused FFN's interleaved stream (x3_ffn_pack), fp16 halves."""
    out = torch.zeros(2 * (wo.numel() + w1.numel() + w2.numel()), dtype=torch.int16, device=wo.device)
    x3_strip_pack(wo, 2, False, 0, out[:2 * wo.numel()])
    x3_strip_pack(w1, 2, False, 1, out[2 * wo.numel():], 4, 0)
    x3_strip_pack(w2, 2, False, 0, out[2 * wo.numel():], 4, 2)
    return out


def attn_out_ffn_fwd(ctx, wpack, bo, residual, gamma1, beta1, p, b1, b2, gamma2, beta2, drop_p=0.0, site_a=0, site_h=0, site_o=0, seed=0, res_mod=0,
                     save=True, hidden_bf16=False, pre_bf16=False):
    """hftt_attn_out_ffn_fwd (x3): x1 = LN1(residual + drop(ctx @ Wo.T + bo)); y = LN2(x1 + drop(FFN(x1))) as ONE launch.
    save=True (training plan) -> (y, x1, pre1, mean1, rstd1, hidden, pre2, mean2, rstd2); save=False (inference plan) -> y only, x1 is never written."""
    _need_cuda(ctx, wpack, residual)
    M, dm = ctx.shape
    dev = ctx.device
    y = torch.empty(M, dm, device=dev)
    mean1 = torch.empty(M, device=dev); rstd1 = torch.empty(M, device=dev); mean2 = torch.empty(M, device=dev); rstd2 = torch.empty(M, device=dev)
    x1 = torch.empty(M, dm, device=dev) if save else None
    pre1 = torch.empty(M, dm, device=dev, dtype=BF16 if pre_bf16 else torch.float32) if save else None
    pre2 = torch.empty(M, dm, device=dev, dtype=BF16 if pre_bf16 else torch.float32) if save else None
    hid = torch.empty(M, p, device=dev, dtype=BF16 if hidden_bf16 else torch.float32) if save else None
    o = StripDesc()
    o.M, o.N, o.K = M, dm, dm
    o.flags = SL_X3_F16 | (SL_PRE_BF16 if pre_bf16 else 0)
    o.x, o.ldx, o.w, o.bias = ctx.data_ptr(), ctx.stride(0), wpack.data_ptr(), bo.data_ptr()
    o.C, o.ldc, o.out_scale, o.gate_scale = (x1.data_ptr() if save else 0), dm, 1.0, 1.0
    o.drop_p, o.drop_site, o.drop_seed = drop_p, site_a, seed
    o.residual, o.ldr, o.res_mod = residual.data_ptr(), residual.stride(0), res_mod
    o.ln_gamma, o.ln_beta = gamma1.data_ptr(), beta1.data_ptr()
    o.pre_ln_out, o.ln_mean, o.ln_rstd = (pre1.data_ptr() if save else 0), mean1.data_ptr(), rstd1.data_ptr()
    f = FfnDesc()
    f.M, f.d, f.p, f.flags, f.mode = M, dm, p, SL_X3_F16 | (SL_H_BF16 if hidden_bf16 else 0) | (SL_PRE_BF16 if pre_bf16 else 0), 0
    f.x, f.ldx, f.w = (x1.data_ptr() if save else y.data_ptr()), dm, wpack.data_ptr() + 2 * (2 * dm * dm)      # (f.x is not read: the strip comes from the registers)
    f.b1, f.b2 = b1.data_ptr(), b2.data_ptr()
    if save:
        f.h_out, f.ldh = hid.data_ptr(), p
    f.drop_p, f.site_h, f.site_o, f.drop_seed = drop_p, site_h, site_o, seed
    f.ln_gamma, f.ln_beta = gamma2.data_ptr(), beta2.data_ptr()
    f.pre_ln_out = pre2.data_ptr() if save else 0
    f.ln_mean, f.ln_rstd = mean2.data_ptr(), rstd2.data_ptr()
    f.y, f.ldy = y.data_ptr(), dm
    check(lib().hftt_attn_out_ffn_fwd(C.byref(o), C.byref(f), _stream(dev)), 'attn_out_ffn_fwd')
    return (y, x1, pre1, mean1, rstd1, hid, pre2, mean2, rstd2) if save else y


def ffn_bwd_dx(dy, wpack_bwd, p, hidden, gate_scale=1.0, residual=None, x3=False, grad_hi=False, dy_drop=None):
    """dh = (hidden > 0) * (dy @ fc_2.weight) * gate_scale;  dx = dh @ fc_1.weight (+ residual) -> (dx, dh); all bf16 (x3: all fp32).
    dy_drop = (p, site, seed) (x3): dy is the gradient of the block's OUTPUT dropout and is masked with that site while it is loaded."""
    _need_cuda(dy, wpack_bwd, hidden)
    M, dm = dy.shape
    dt = torch.float32 if x3 else BF16
    dx = torch.empty(M, dm, device=dy.device, dtype=dt)
    hbf = x3 and hidden.dtype == BF16                     # x3 with the hidden stored as bf16: dh leaves as bf16 too (SL_H_BF16)
    dh = torch.empty(M, p, device=dy.device, dtype=BF16 if hbf else dt)
    d = FfnDesc()
    d.M, d.d, d.p, d.flags, d.mode = M, dm, p, ((SL_X3_BF16 | (SL_H_BF16 if hbf else 0) | (SL_X3_GRAD_HI if grad_hi else 0)) if x3 else (SL_X_BF16 | SL_C_BF16 | SL_RES_BF16)), 1
    d.x, d.ldx, d.w = dy.data_ptr(), dy.stride(0), wpack_bwd.data_ptr()
    d.h_out, d.ldh = dh.data_ptr(), p
    d.gate, d.ldg, d.gate_scale = hidden.data_ptr(), hidden.stride(0), gate_scale
    if dy_drop is not None:
        d.drop_p, d.site_o, d.drop_seed = dy_drop
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
    d.y, d.ldy = dx.data_ptr(), dm
    check(lib().hftt_ffn_bwd_dx(C.byref(d), _stream(dy.device)), 'ffn_bwd_dx')
    return dx, dh


def gemm_tn(dY, X, npass=3, out_scale=1.0, with_bias=True, grad_hi=False, dy_drop=None):
    """dW[N,K] = out_scale * dY[M,N].T @ X[M,K]; db[N] = colsum(dY).
    dy_drop = (p, site, seed) (npass 4, fp32 dY): HFTT_TN_DY_DROP -- dY is masked with that dropout site while it is loaded."""
    _need_cuda(dY, X)
    M, N = dY.shape
    K = X.shape[1]
    dW = torch.empty(N, K, device=dY.device)
    db = torch.empty(N, device=dY.device)
    L = lib()
    wsb = L.hftt_gemm_tn_ws_bytes(M, N, K)
    ws = torch.empty(wsb // 4 + 16, device=dY.device)
    d = GemmTnDesc()
    d.M, d.N, d.K, d.npass = M, N, K, npass
    d.dY, d.lddy, d.X, d.ldx = dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0)
    d.out_scale, d.beta, d.n_seg = out_scale, 0.0, 1
    d.io_flags = (TN_DY_BF16 if dY.dtype == BF16 else 0) | (TN_X_BF16 if X.dtype == BF16 else 0) | (TN_DY_HI if (grad_hi and dY.dtype != BF16) else 0) \
        | (TN_DY_DROP if dy_drop is not None else 0)
    if dy_drop is not None:
        d.drop_p, d.drop_site, d.drop_seed = dy_drop
    d.seg_row0[0], d.seg_rows[0], d.seg_dw[0], d.seg_db[0] = 0, N, dW.data_ptr(), (db.data_ptr() if with_bias else 0)
    d.K_out = K
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    check(L.hftt_gemm_tn(C.byref(d), _stream(dY.device)), 'gemm_tn')
    return dW, db


def to_planes(x):
    """fp32 [..., cols] (contiguous, cols % 32 == 0) -> the f16-pair plane form of the same shape (hftt_x3_to_planes): per 32-column group the
    32 fp16 hi halves, then the 32 lo halves, in the group's 128 bytes.  A float32 tensor whose BYTES are the planes."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 32 == 0
    out = torch.empty_like(x)
    cols = x.shape[-1]
    check(lib().hftt_x3_to_planes(x.data_ptr(), cols, out.data_ptr(), cols, x.numel() // cols, cols, _stream(x.device)), 'x3_to_planes')
    return out


def _attn_desc(q, k, v, n_heads, npass, drop_p, drop_site, drop_seed, planes=False):
    n_seq, Lq, dm = q.shape
    Lk = k.shape[1]
    d = AttnDesc()
    d.n_seq, d.n_heads, d.Lq, d.Lk, d.dh, d.npass = n_seq, n_heads, Lq, Lk, dm // n_heads, npass
    d.q, d.q_seq_stride, d.ldq = q.data_ptr(), q.stride(0), q.stride(1)
    d.k, d.k_seq_stride, d.ldk = k.data_ptr(), k.stride(0), k.stride(1)
    d.v, d.v_seq_stride, d.ldv = v.data_ptr(), v.stride(0), v.stride(1)
    d.drop_p, d.drop_site, d.drop_seed = drop_p, drop_site, drop_seed
    d.io_flags = (ATTN_Q_BF16 if q.dtype == BF16 else 0) | (ATTN_KV_BF16 if k.dtype == BF16 else 0) | ((ATTN_Q_F16PAIR | ATTN_KV_F16PAIR) if planes else 0)
    assert k.dtype == v.dtype
    return d


def attn_fwd(q, k, v, n_heads, npass=3, want_probs=False, drop_p=0.0, drop_site=0, drop_seed=0, out_dtype=torch.float32, planes=False):
    """q [n_seq, Lq, d], k/v [n_seq, Lk, d] (any row/seq strides) -> out [n_seq, Lq, d], row stats [n_seq, H, Lq, 2](, probs).
    planes (npass 2): q, k, v hold f16-pair planes (to_planes / a projection with SL_C_F16PAIR) instead of fp32 values."""
    _need_cuda(q, k, v)
    n_seq, Lq, dm = q.shape
    Lk = k.shape[1]
    out = torch.empty(n_seq, Lq, dm, device=q.device, dtype=out_dtype)
    lse = torch.empty(n_seq, n_heads, Lq, 2, device=q.device)     # (row max, 1/row sum)
    probs = torch.empty(n_seq, n_heads, Lq, Lk, device=q.device) if want_probs else None
    d = _attn_desc(q, k, v, n_heads, npass, drop_p, drop_site, drop_seed, planes)
    d.out, d.o_seq_stride, d.ldo = out.data_ptr(), out.stride(0), out.stride(1)
    d.lse = lse.data_ptr()
    d.probs = probs.data_ptr() if want_probs else 0
    d.io_flags |= ATTN_O_BF16 if out_dtype == BF16 else 0
    check(lib().hftt_attn_fwd(C.byref(d), _stream(q.device)), 'attn_fwd')
    return (out, lse, probs) if want_probs else (out, lse)


def attn_bwd(q, k, v, out, lse, dout, n_heads, npass=3, drop_p=0.0, drop_site=0, drop_seed=0, dq_dtype=torch.float32, dkv_dtype=torch.float32,
             grads_out=None, planes=False):
    """grads_out = (dq, dk, dv): preallocated (possibly strided, e.g. interleaved [S, 3d]) gradient tensors to write into."""
    _need_cuda(q, k, v, out, dout)
    if grads_out is not None:
        dq, dk, dv = grads_out
        dq_dtype, dkv_dtype = dq.dtype, dk.dtype
    else:
        dq = torch.empty(q.shape, device=q.device, dtype=dq_dtype)
        dk = torch.empty(k.shape, device=q.device, dtype=dkv_dtype)
        dv = torch.empty(v.shape, device=q.device, dtype=dkv_dtype)
    d = _attn_desc(q, k, v, n_heads, npass, drop_p, drop_site, drop_seed, planes)
    d.out, d.o_seq_stride, d.ldo = out.data_ptr(), out.stride(0), out.stride(1)
    d.lse = lse.data_ptr()
    assert dout.stride() == out.stride() and dout.dtype == out.dtype
    d.dout = dout.data_ptr()
    d.io_flags |= (ATTN_O_BF16 if out.dtype == BF16 else 0) | (ATTN_DQ_BF16 if dq_dtype == BF16 else 0) | (ATTN_DKV_BF16 if dkv_dtype == BF16 else 0)
    d.dq, d.dq_seq_stride, d.lddq = dq.data_ptr(), dq.stride(0), dq.stride(1)
    d.dk, d.dk_seq_stride, d.lddk = dk.data_ptr(), dk.stride(0), dk.stride(1)
    d.dv, d.dv_seq_stride, d.lddv = dv.data_ptr(), dv.stride(0), dv.stride(1)
    check(lib().hftt_attn_bwd(C.byref(d), _stream(q.device)), 'attn_bwd')
    return dq, dk, dv


def ln_bwd(dy, r, mean, rstd, gamma, drop_p=0.0, drop_site=0, drop_seed=0, drop_dtype=torch.float32, dr_dtype=torch.float32):
    """dy may be stored as bf16 (bf16 gradient stream); dr is written as dr_dtype."""
    _need_cuda(dy, r)
    M, N = dy.shape
    L = lib()
    n_wg = L.hftt_ln_bwd_wgs(M)
    ws = torch.empty(n_wg * 2 * N, device=dy.device)
    dr = torch.empty(dy.shape, device=dy.device, dtype=dr_dtype)
    drd = torch.empty(dy.shape, device=dy.device, dtype=drop_dtype) if drop_p > 0 else None
    d = LnBwdDesc()
    d.M, d.N = M, N
    d.dy, d.r, d.mean, d.rstd, d.gamma = dy.data_ptr(), r.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr()
    d.dr, d.dr_drop = dr.data_ptr(), (drd.data_ptr() if drd is not None else 0)
    d.drop_p, d.drop_site, d.drop_seed = drop_p, drop_site, drop_seed
    d.ws = ws.data_ptr()
    d.drop_bf16 = 1 if drop_dtype == BF16 else 0
    d.io_flags = (1 if dy.dtype == BF16 else 0) | (2 if dr_dtype == BF16 else 0) | (4 if r.dtype == BF16 else 0)
    if r.dtype not in (BF16, torch.float32) or dy.dtype not in (BF16, torch.float32):
        raise HfttError('ln_bwd: dy / r must be fp32 or bf16')
    st = _stream(dy.device)
    check(L.hftt_ln_bwd(C.byref(d), st), 'ln_bwd')
    dg = torch.empty(N, device=dy.device)
    db = torch.empty(N, device=dy.device)
    check(L.hftt_ln_bwd_reduce(ws.data_ptr(), n_wg, N, dg.data_ptr(), db.data_ptr(), 0.0, st), 'ln_bwd_reduce')
    return dr, drd, dg, db


def colsum(x, beta=0.0, out=None):
    _need_cuda(x)
    rows, n = x.shape
    L = lib()
    ws = torch.empty(L.hftt_colsum_ws_bytes(rows, n) // 4 + 16, device=x.device)
    if out is None:
        out = torch.zeros(n, device=x.device)
    check(L.hftt_colsum(x.data_ptr(), rows, n, x.stride(0), out.data_ptr(), beta, ws.data_ptr(), 1 if x.dtype == BF16 else 0, _stream(x.device)), 'colsum')
    return out


def adam_step(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _need_cuda(p, g, m, v)
    check(lib().hftt_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), step, lr, beta1, beta2, eps,
                               grad_scale, _stream(p.device)), 'adam_step')


# ------------------------------------------------------------------------------------------------
# log-mel front end (model/amt.py:55-63)
# ------------------------------------------------------------------------------------------------
class LogMel:
    """Device-resident tables (hann window, FFT twiddles, sparse slaney/htk mel filterbank) + the kernel launch."""

    def __init__(self, device, sr=16000, n_fft=2048, hop=256, n_mels=256, log_offset=1e-8):
        self.device = torch.device(device)
        _need_cuda(torch.empty(0, device=self.device))
        self.sr, self.n_fft, self.hop, self.n_mels, self.log_offset = sr, n_fft, hop, n_mels, log_offset
        n = np.arange(n_fft, dtype=np.float64)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)                       # periodic hann
        k = np.arange(n_fft // 2, dtype=np.float64)
        tw = np.concatenate([np.cos(2.0 * np.pi * k / n_fft), np.sin(2.0 * np.pi * k / n_fft)])
        fb = self.mel_filterbank(sr, n_fft, n_mels)                                # [n_freqs, n_mels] float64
        starts, lens, offs, weights = [], [], [], []
        for m in range(n_mels):
            nz = np.nonzero(fb[:, m])[0]
            if len(nz) == 0:
                starts.append(0); lens.append(0); offs.append(len(weights)); continue
            s, e = int(nz[0]), int(nz[-1]) + 1
            starts.append(s); lens.append(e - s); offs.append(len(weights))
            weights.extend(fb[s:e, m].tolist())
        self.nnz = len(weights)
        dev = self.device
        self.window = torch.tensor(window, dtype=torch.float32, device=dev)
        self.twiddle = torch.tensor(tw, dtype=torch.float32, device=dev)
        self.fb_start = torch.tensor(starts, dtype=torch.int32, device=dev)
        self.fb_len = torch.tensor(lens, dtype=torch.int32, device=dev)
        self.fb_off = torch.tensor(offs, dtype=torch.int32, device=dev)
        self.fb_w = torch.tensor(weights if weights else [0.0], dtype=torch.float32, device=dev)

    @staticmethod
    def mel_filterbank(sr, n_fft, n_mels):
        """htk mel points, triangles on linspace(0, sr/2, n_fft/2+1), slaney area normalisation (float32 steps as
        torchaudio.functional.melscale_fbanks computes them)."""
        n_freqs = n_fft // 2 + 1
        all_freqs = np.linspace(0, sr // 2, n_freqs, dtype=np.float32)
        m_max = 2595.0 * math.log10(1.0 + (sr / 2.0) / 700.0)
        m_pts = np.linspace(0.0, m_max, n_mels + 2, dtype=np.float32)
        f_pts = (700.0 * (np.power(np.float32(10.0), m_pts / np.float32(2595.0)) - 1.0)).astype(np.float32)
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts[None, :] - all_freqs[:, None]
        down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
        up = slopes[:, 2:] / f_diff[1:]
        fb = np.maximum(0.0, np.minimum(down, up))
        enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
        return (fb * enorm[None, :]).astype(np.float64)

    def __call__(self, wave_mono):
        """wave [n_samples] fp32 at self.sr (device tensor) -> log-mel [n_frames, n_mels], n_frames = 1 + n//hop."""
        _need_cuda(wave_mono)
        wave = wave_mono.contiguous().float()
        n = wave.numel()
        n_frames = 1 + n // self.hop
        feat = torch.empty(n_frames, self.n_mels, device=self.device)
        d = LogmelDesc()
        d.wave, d.n_samples = wave.data_ptr(), n
        d.n_fft, d.hop, d.n_mels, d.n_frames = self.n_fft, self.hop, self.n_mels, n_frames
        d.window, d.twiddle = self.window.data_ptr(), self.twiddle.data_ptr()
        d.fb_start, d.fb_len, d.fb_off, d.fb_w = self.fb_start.data_ptr(), self.fb_len.data_ptr(), self.fb_off.data_ptr(), self.fb_w.data_ptr()
        d.log_offset = self.log_offset
        d.feat = feat.data_ptr()
        check(lib().hftt_logmel(C.byref(d), _stream(self.device)), 'logmel')
        return feat


def resample_kernel_table(sr_in, sr_out, lowpass_filter_width=6, rolloff=0.99):
    """(kernel [up, taps] float64 tensor, up, down, width) of torchaudio.transforms.Resample's defaults (Hann-windowed sinc): the published
    algorithm restated (torchaudio is absent: parity unpinned at that boundary)."""
    g = math.gcd(int(sr_in), int(sr_out))
    down, up = int(sr_in) // g, int(sr_out) // g
    base = min(down, up) * rolloff
    width = math.ceil(lowpass_filter_width * down / base)
    idx = torch.arange(-width, width + down, dtype=torch.float64)[None, :] / down
    t = (torch.arange(0, -up, -1, dtype=torch.float64)[:, None] / up + idx) * base
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / down)
    return kern, up, down, width


_RESAMPLE_TABLES = {}


def resample(wave_mono, sr_in, sr_out):
    """wave [n] fp32 device tensor at sr_in -> [ceil(n * sr_out / sr_in)] at sr_out (hftt_resample: polyphase, fp32)."""
    _need_cuda(wave_mono)
    wave = wave_mono.contiguous().float()
    key = (int(sr_in), int(sr_out), str(wave.device))
    if key not in _RESAMPLE_TABLES:                  # (the table is a pure function of the two rates: built and uploaded once per device)
        kern, up, down, width = resample_kernel_table(sr_in, sr_out)
        _RESAMPLE_TABLES[key] = (kern.float().contiguous().to(wave.device), up, down, width)
    kd, up, down, width = _RESAMPLE_TABLES[key]
    n = wave.numel()
    n_out = -(-n * up // down)
    out = torch.empty(n_out, device=wave.device)
    d = ResampleDesc()
    d.wave, d.n_in, d.kernel = wave.data_ptr(), n, kd.data_ptr()
    d.up, d.down, d.width, d.taps = up, down, width, kd.shape[1]
    d.out, d.n_out = out.data_ptr(), n_out
    check(lib().hftt_resample(C.byref(d), _stream(wave.device)), 'resample')
    return out
