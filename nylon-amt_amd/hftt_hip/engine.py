"""Host-side engine of the MI355X hFT-Transformer path.

It owns (a) ONE flat fp32 parameter buffer + ONE flat gradient buffer (the module's nn.Parameters are views
into the first), (b) the prepared bf16 weight planes the MFMA kernels consume, (c) all activation /
gradient workspaces for a batch size, and (d) pre-built launch plans: lists of C-ABI calls
(libhftt_hip.so) with their descriptors filled in once, replayed every step on torch's current HIP stream.

PyTorch is used for device memory, streams and autograd glue only; every arithmetic op of the path is a
HIP kernel behind the C ABI (include/hftt_hip.h).  There is no fallback: without the library or on a CPU
tensor the engine raises.

Reference algorithm: hftt_code/model/model_spec2midi.py (forward), training/train.py:141-159 (loss,
backward, step).  Shapes follow the reference: T frames, F bins, N notes, V velocities, d hidden, p ffn.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict

import torch

from . import _capi
from ._capi import (AttnDesc, FfnDesc, FoldDesc, GemmNtDesc, GemmTnDesc, LnBwdDesc, LossDesc, PrepEntry, StripDesc, StripPackEntry,
                    SL_C_BF16, SL_C_F16PAIR, SL_H_BF16, SL_PRE_BF16, SL_RELU, SL_X3_GRAD_HI, SL_RES_BF16, SL_X_BF16, SL_X3_F16, SL_X3_BF16, SL_X_DROP,
                    ATTN_Q_F16PAIR, ATTN_KV_F16PAIR, check, lib)

# precision -> the descriptors' `npass` code (include/hftt_hip.h): 'x3' = split fp16 on forward products (2) and split bf16 on products with a
# gradient operand (4): three bf16-rate MFMA passes per product, fp32 tensors in HBM, outputs within 1e-3 of the reference (measured 1e-4)
PRECISION_NPASS = {'parity': 3, 'bf16': 1, 'x3': 2}


def keep_scale(p):
    """scale of the kept elements of a dropout site (csrc/hftt_common.h: hftt_keep_scale): 256 / thr, thr = round((1 - p) * 256) -- the
    reciprocal of the keep probability the 8-bit generator actually applies, so E[dropout(x)] = x exactly (nn.Dropout's contract)."""
    if not p > 0.0:
        return 1.0
    import numpy as np
    kk = (1.0 - float(np.float32(p))) * 256.0 + 0.5
    thr = 256 if kk >= 256.0 else (0 if kk <= 0.0 else int(kk))
    return float(np.float32(256.0) / np.float32(thr)) if thr > 0 else 0.0


def _align(x, a):
    return (x + a - 1) // a * a


class _Flat:
    """Bump allocator over one flat tensor (element offsets, 16-byte aligned)."""

    def __init__(self):
        self.off = 0
        self.items = OrderedDict()

    def add(self, name, numel, align=8):
        self.off = _align(self.off, align)
        self.items[name] = self.off
        self.off += numel
        return self.items[name]


class HfttEngine:
    _dropout_notice_given = False

    def __init__(self, cfg, device, precision='parity', dropout=0.0, seed=1234):
        """cfg: dict with n_margin,n_frame,n_bin,cnn_channel,cnn_kernel,hid_dim,pf_dim,enc_layer,dec_layer,
        enc_head,dec_head,n_note,n_velocity (the constructor arguments of the reference classes)."""
        self.lib = lib()
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _capi.HfttError('HfttEngine needs a ROCm device (got %s): the HIP kernels are the only compute path' % device)
        c = self.cfg
        self.T, self.F, self.N, self.V = c['n_frame'], c['n_bin'], c['n_note'], c['n_velocity']
        self.d, self.p = c['hid_dim'], c['pf_dim']
        self.He, self.Hd = c['enc_head'], c['dec_head']
        self.Le, self.Ld = c['enc_layer'], c['dec_layer']
        self.n_proc = 2 * c['n_margin'] + 1
        self.Kp = _align(self.n_proc, 32)
        self.W = self.T + 2 * c['n_margin']
        self.nw = self.n_proc - (c['cnn_kernel'] - 1)
        if self.d % 32 or self.p % 32 or self.d not in (64, 128, 256):
            raise _capi.HfttError('hid_dim must be 64/128/256 and pf_dim a multiple of 32 (got %d/%d)' % (self.d, self.p))
        for h in (self.He, self.Hd):
            if self.d % h or (self.d // h) not in (32, 64):
                raise _capi.HfttError('head_dim must be 32 or 64 (hid_dim %d, heads %d)' % (self.d, h))
        if max(self.T, self.F, self.N) > 256:
            raise _capi.HfttError('sequence axes must be <= 256')
        if self.V % 4:
            raise _capi.HfttError('n_velocity must be a multiple of 4')
        self.NH = self.V + 3                       # packed head rows: velocity[0:V], onset, offset, mpe
        self.NHp = _align(self.NH + 1, 64)
        self.store_bf16_opt = os.environ.get('HFTT_BF16_STORE', '1') != '0'
        self.strip_opt = os.environ.get('HFTT_STRIP', '1') != '0'
        self.planes_opt = os.environ.get('HFTT_X3_PLANES', '1') != '0'      # x3 strip plans: q / k / v between projection and attention as f16-pair planes
        # x3 strip plans: the cross-attention K / V projections of ALL decoder layers (the same input: the encoder output, model_spec2midi.py:259,296)
        # as ONE launch with N = Ld * 2d instead of Ld launches that each re-read the encoder output
        self.merge_ckv_opt = os.environ.get('HFTT_X3_MERGE_CKV', '1') != '0'
        # x3 strip plans (d = 256): the LayerNorm backward writes NO dropout-masked copy of its result; its consumers -- the weight-gradient
        # product (HFTT_TN_DY_DROP), the fused FFN's dX (site_o) and the fc_o dX (HFTT_SL_X_DROP) -- apply the mask while they load dr
        self.ln_mask_in_consumers_opt = os.environ.get('HFTT_X3_LN_MASK_IN_CONSUMERS', '1') != '0'
        # round 6 (ABI v8): fc_o + residual + LayerNorm and the FFN block behind it as ONE launch (hftt_attn_out_ffn_fwd) -- 'all' in both forward plans,
        # 'inference' only in the plan that writes nothing in between, '0' never
        self.fuse_offn_opt = os.environ.get('HFTT_X3_FUSE_OFFN', 'all')
        self.set_precision(precision)
        self.dropout = float(dropout)
        # The device generator decides per element with ONE byte of a hash word (csrc/hftt_common.h: hftt_keep_thr), so the keep probability is
        # quantised to 1/256: `-dropout 0.1` of m_training.py is applied as 26/256 = 0.1016 (kept elements are scaled by 256/230, the reciprocal of
        # the probability actually applied: E[dropout(x)] = x).  Not silent: the applied rate is an attribute and a one-time notice.
        self.dropout_applied = (1.0 - 1.0 / keep_scale(self.dropout)) if self.dropout > 0.0 else 0.0
        if self.dropout > 0.0 and abs(self.dropout_applied - self.dropout) > 1e-6 and not HfttEngine._dropout_notice_given:
            HfttEngine._dropout_notice_given = True
            import warnings
            warnings.warn('hftt_hip: dropout p = %g is applied as %d/256 = %.4f (8-bit keep decisions of the device generator; kept elements are '
                          'scaled by the reciprocal of the applied keep probability, so the expectation is unchanged)'
                          % (self.dropout, round(self.dropout_applied * 256), self.dropout_applied), stacklevel=2)
        self.base_seed = int(seed)
        self.step_counter = 0
        self.generation = 0
        self._bound = None                          # list of (name, param, offset, numel)
        self._ws = {}
        self._site = 0
        self.profiler = None                        # optional per-launch HIP-event timer (bench.py)
        self._in_backward = False                   # set while the backward plan is being built (x3: products with a gradient operand)
        self.frozen_weights = False                 # the caller's promise that the parameters do not change (see prepare_weights)
        self._prepared_frozen = False

    # ------------------------------------------------------------------ precision / parameters
    def set_precision(self, precision):
        if precision not in PRECISION_NPASS:
            raise ValueError('precision must be one of %s' % list(PRECISION_NPASS))
        self.precision = precision
        self.npass = PRECISION_NPASS[precision]
        self._ws = {}
        # bf16 mode: tensors consumed only as MFMA operands (projections, attention context, FFN hidden and their gradients)
        # are STORED as bf16 -- identical numerics (they were rounded at load time anyway), half the traffic
        self.sb = (self.npass == 1) and getattr(self, 'store_bf16_opt', True)
        # strip kernels (csrc/strip_gemm.hip): bf16 mode at the paper's width.  Then the WHOLE activation stream between kernels is
        # bf16 (residual stream, pre-LayerNorm sums, hidden), fp32 lives only inside a kernel (accumulators, LayerNorm statistics).
        self.strip = getattr(self, 'strip_opt', True) and self.d == 256 and ((self.sb and self.p % 64 == 0) or (self.npass == 2 and self.p == 512))
        # the reference's default width (training/m_training.py:56-61: d = 64, ff = 128) in the x3 mode: the same launch sequence on the
        # small-width strip family (csrc/x3s_strip.h: every weight matrix of a launch resident in LDS, compact packs)
        # ... and, since round 5, in the bf16 mode (csrc/bs_strip.hip: the same launch sequence on the bf16 stream -- BASELINE config 2; the packs
        # are the x3 family's compact ones with bf16 halves, of which the bf16 kernels read the hi fragments)
        self.strip_small = getattr(self, 'strip_opt', True) and (self.npass == 2 or self.sb) and self.d == 64 and self.p == 128
        self.strip = self.strip or self.strip_small
        # bfs: the bf16 activation / gradient STREAM of the bf16 strip plans.  The x3 strip plans run the same launch sequence on fp32 tensors.
        self.bfs = self.strip and self.sb
        self.x3 = self.npass == 2
        # x3 strip plans: the STORED copy of the FFN hidden and its gradient dh are bf16.  fc_2 takes the hidden from registers at full width;
        # the stored copy is read as the ReLU / dropout gate and as ONE factor of the weight-gradient products (8 mantissa bits of one factor
        # leave dW's direction untouched: tests/test_paper_bf16_gpu.py), and these two tensors were 2 x 537 MB per layer at S_e.
        self.hh = self.strip and self.x3 and os.environ.get('HFTT_X3_FP32_HIDDEN', '0') != '1'
        # Option (HFTT_X3_GRAD_HI=1): the GRADIENT operand of every GEMM-shaped backward product (dY of dW = dY^T X, the strip of a dX kernel,
        # A of the block dX GEMMs) enters as its bf16 rounding only -- two MFMA passes against the saved operand's / the weights' bf16 pair
        # instead of three: +3.5 % (260 against 251 clips/s on one box), every gradient tensor's cosine against the exact-fp32 mode still
        # >= 0.9999 at paper size, but the gradients move from 2e-4 to 3e-3 .. 6e-3 of the fp32 reference's (relative to the tensor's maximum) --
        # outside the 1e-3 the default mode keeps for gradients too, hence opt-in.
        self.g8 = self.x3 and os.environ.get('HFTT_X3_GRAD_HI', '0') == '1' and not self.strip_small
        if self.g8 and not (lib().hftt_build_options() & 1):
            raise HfttError('HFTT_X3_GRAD_HI=1 needs a library built with HFTT_BUILD_GRAD_HI=1 (python nylon-amt_amd/build.py writes '
                            'libhftt_hip_g.so; hftt_hip/_capi.py loads it under the same switch): the default library does not carry the '
                            'gradient-rounding kernels')
        self._prepared_frozen = False
        if getattr(self, '_bound', None) is not None:
            self._build_prep()

    def bind(self, named_params):
        """Flatten the module's parameters into one buffer (reference state_dict order) and rebind .data as views."""
        named = list(named_params)
        lay = _Flat()
        offs = [lay.add(n, p.numel()) for n, p in named]
        total = _align(lay.off, 8)
        flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        for (n, p), o in zip(named, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).to(self.device, torch.float32))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self.flat_params = flat
        self.flat_grads = torch.zeros_like(flat)
        self.poff = {n: o for (n, _), o in zip(named, offs)}
        self.pshape = {n: tuple(p.shape) for n, p in named}
        self._bound = [(n, p, o, p.numel()) for (n, p), o in zip(named, offs)]
        from .trainer import register_binding     # lets FusedAdam(model.parameters()) built before this bind find the engine
        register_binding(self, [p for _, p in named])
        self._build_prep()
        self._ws = {}

    def is_bound(self):
        if self._bound is None:
            return False
        base = self.flat_params.data_ptr()
        for _, p, o, _ in self._bound:
            if p.data_ptr() != base + 4 * o:
                return False
        return True

    def grad_views(self, source=None):
        src = self.flat_grads if source is None else source
        return [src[o:o + n].view(self.pshape[name]) for name, _, o, n in self._bound]

    def P(self, name):   # device address of a parameter
        return self.flat_params.data_ptr() + 4 * self.poff[name]

    def G(self, name):   # device address of a parameter's gradient
        return self.flat_grads.data_ptr() + 4 * self.poff[name]

    # ------------------------------------------------------------------ prepared weights
    def _build_prep(self):
        d, p = self.d, self.p
        wl, fl = _Flat(), _Flat()
        entries = []

        def mat(key, srcs, rows_each, cols, transposed):
            """srcs: parameter names stacked along rows; returns plane offset of the [sum rows, cols] (or transposed) matrix."""
            rows = rows_each * len(srcs)
            if transposed:
                off = wl.add(key, cols * _align(rows, 32), 64)
                for i, s in enumerate(srcs):
                    entries.append((self.poff[s], off + i * rows_each, rows_each, cols, cols, _align(rows, 32), 1))
            else:
                off = wl.add(key, _align(rows, 64) * cols, 64)
                for i, s in enumerate(srcs):
                    entries.append((self.poff[s], off + i * rows_each * cols, rows_each, cols, cols, cols, 0))
            return off

        def vec(key, srcs, n_each, pad_to=None):
            n = n_each * len(srcs)
            off = fl.add(key, pad_to or n, 8)
            for i, s in enumerate(srcs):
                entries.append((self.poff[s], off + i * n_each, 1, n_each, n_each, n_each, 2))
            return off

        W = {}

        def attn_self(pre, key):
            names = [pre + 'fc_q', pre + 'fc_k', pre + 'fc_v']
            W[key + '.qkv'] = mat(key + '.qkv', [n + '.weight' for n in names], d, d, False)
            W[key + '.qkv_t'] = mat(key + '.qkv_t', [n + '.weight' for n in names], d, d, True)
            W[key + '.qkv_b'] = vec(key + '.qkv_b', [n + '.bias' for n in names], d)
            W[key + '.o'] = mat(key + '.o', [pre + 'fc_o.weight'], d, d, False)
            W[key + '.o_t'] = mat(key + '.o_t', [pre + 'fc_o.weight'], d, d, True)

        def attn_cross(pre, key):
            W[key + '.q'] = mat(key + '.q', [pre + 'fc_q.weight'], d, d, False)
            W[key + '.q_t'] = mat(key + '.q_t', [pre + 'fc_q.weight'], d, d, True)
            names = [pre + 'fc_k', pre + 'fc_v']
            W[key + '.kv'] = mat(key + '.kv', [n + '.weight' for n in names], d, d, False)
            W[key + '.kv_t'] = mat(key + '.kv_t', [n + '.weight' for n in names], d, d, True)
            W[key + '.kv_b'] = vec(key + '.kv_b', [n + '.bias' for n in names], d)
            W[key + '.o'] = mat(key + '.o', [pre + 'fc_o.weight'], d, d, False)
            W[key + '.o_t'] = mat(key + '.o_t', [pre + 'fc_o.weight'], d, d, True)

        def ffn(pre, key):
            # fc_1.weight [p, d], fc_2.weight [d, p]
            off = wl.add(key + '.f1', _align(p, 64) * d, 64); entries.append((self.poff[pre + 'fc_1.weight'], off, p, d, d, d, 0)); W[key + '.f1'] = off
            off = wl.add(key + '.f1_t', d * p, 64); entries.append((self.poff[pre + 'fc_1.weight'], off, p, d, d, p, 1)); W[key + '.f1_t'] = off
            off = wl.add(key + '.f2', d * p, 64); entries.append((self.poff[pre + 'fc_2.weight'], off, d, p, p, p, 0)); W[key + '.f2'] = off
            off = wl.add(key + '.f2_t', _align(p, 64) * d, 64); entries.append((self.poff[pre + 'fc_2.weight'], off, d, p, p, d, 1)); W[key + '.f2_t'] = off

        def heads(tag, key):
            pre = 'decoder_spec2midi.'
            NHp, V = self.NHp, self.V
            off = wl.add(key, NHp * d, 64)
            W[key] = off
            entries.append((self.poff[f'{pre}fc_velocity_{tag}.weight'], off, V, d, d, d, 0))
            for i, nm in enumerate(('onset', 'offset', 'mpe')):
                entries.append((self.poff[f'{pre}fc_{nm}_{tag}.weight'], off + (V + i) * d, 1, d, d, d, 0))
            offt = wl.add(key + '_t', d * NHp, 64)
            W[key + '_t'] = offt
            entries.append((self.poff[f'{pre}fc_velocity_{tag}.weight'], offt, V, d, d, NHp, 1))
            for i, nm in enumerate(('onset', 'offset', 'mpe')):
                entries.append((self.poff[f'{pre}fc_{nm}_{tag}.weight'], offt + V + i, 1, d, d, NHp, 1))
            ob = fl.add(key + '_b', NHp, 8)
            W[key + '_b'] = ob
            entries.append((self.poff[f'{pre}fc_velocity_{tag}.bias'], ob, 1, V, V, V, 2))
            for i, nm in enumerate(('onset', 'offset', 'mpe')):
                entries.append((self.poff[f'{pre}fc_{nm}_{tag}.bias'], ob + V + i, 1, 1, 1, 1, 2))

        sl = _Flat()
        sentries = []

        x3 = self.npass == 2
        sentries_t = []                              # x3: the transposed (backward) matrices are packed as bf16 halves, the others as fp16 halves

        def spack(key, parts, Ktot, transpose=False, order=0, stride=1, offset=0, base=None, numel=None):
            """parts: (parameter name, n0, k0) blocks of the logical [N, Ktot] matrix; returns the stream's element offset.
            x3: every fragment is a (hi, lo) pair -- twice the elements; slots come in pairs (stream position = offset + stride * (slot >> 1) +
            (slot & 1)), so a plain stream has stride 2 and the fused block's two matrices stride 4 with offsets 0 / 2."""
            if x3:
                numel = None if numel is None else 2 * numel
                stride, offset = 2 * stride, 2 * offset
            if base is None:
                base = sl.add(key, numel, 512)
                W['s.' + key] = base
            for name, n0, k0 in parts:
                rows, cols = self.pshape[name]
                (sentries_t if (x3 and transpose) else sentries).append((self.poff[name], base, rows, cols, cols, 1 if transpose else 0, n0, k0, Ktot, order, stride, offset))
            return base

        def spack_s(key, parts, Ktot, Ntot, transpose=False, pair_offset=0, base=None, total_pairs=None):
            """compact pack of the small-width family (order 2): the (hi, lo) pair of (k chunk c, tile t) at pair index pair_offset + c * NT + t,
            1024 int16 elements per pair"""
            if base is None:
                base = sl.add(key, (total_pairs or (Ktot // 16) * (Ntot // 32)) * 1024, 512)
                W['s.' + key] = base
            for name, n0, k0 in parts:
                rows, cols = self.pshape[name]
                (sentries_t if transpose else sentries).append((self.poff[name], base, rows, cols, cols, 1 if transpose else 0, n0, k0, Ktot, 2, Ntot // 32, pair_offset))
            return base

        def strip_attn_s(pre, key, cross):
            wq, wk, wv, wo = (pre + n + '.weight' for n in ('fc_q', 'fc_k', 'fc_v', 'fc_o'))
            if cross:
                spack_s(key + '.q', [(wq, 0, 0)], d, d)
                spack_s(key + '.kv', [(wk, 0, 0), (wv, d, 0)], d, 2 * d)
                spack_s(key + '.q_t', [(wq, 0, 0)], d, d, transpose=True)
                spack_s(key + '.kv_t', [(wk, 0, 0), (wv, 0, d)], 2 * d, d, transpose=True)
            else:
                spack_s(key + '.qkv', [(wq, 0, 0), (wk, d, 0), (wv, 2 * d, 0)], d, 3 * d)
                spack_s(key + '.qkv_t', [(wq, 0, 0), (wk, 0, d), (wv, 0, 2 * d)], 3 * d, d, transpose=True)
            spack_s(key + '.o', [(wo, 0, 0)], d, d)
            spack_s(key + '.o_t', [(wo, 0, 0)], d, d, transpose=True)

        def strip_ffn_s(pre, key):
            w1, w2 = pre + 'fc_1.weight', pre + 'fc_2.weight'          # [p, d], [d, p]
            n1, n2 = (d // 16) * (p // 32), (p // 16) * (d // 32)
            base = spack_s(key + '.ffn', [(w1, 0, 0)], d, p, total_pairs=n1 + n2)
            spack_s(key + '.ffn', [(w2, 0, 0)], p, d, pair_offset=n1, base=base)
            base = spack_s(key + '.ffn_t', [(w2, 0, 0)], d, p, transpose=True, total_pairs=n1 + n2)       # fc_2.weight^T [p, d]
            spack_s(key + '.ffn_t', [(w1, 0, 0)], p, d, transpose=True, pair_offset=n1, base=base)       # fc_1.weight^T [d, p]

        def strip_attn(pre, key, cross):
            if self.strip_small:
                return strip_attn_s(pre, key, cross)
            wq, wk, wv, wo = (pre + n + '.weight' for n in ('fc_q', 'fc_k', 'fc_v', 'fc_o'))
            tm = 1 if x3 else 0                       # x3: the K == 256 linears without LayerNorm take the tile-major pack (csrc/x3_strip.hip)
            if cross:
                spack(key + '.q', [(wq, 0, 0)], d, order=tm, numel=d * d)
                if not getattr(self, 'merge_ckv', False):   # (merged: one stream for all layers, 'dec.ca.kv_all' below)
                    spack(key + '.kv', [(wk, 0, 0), (wv, d, 0)], d, order=tm, numel=2 * d * d)
                spack(key + '.q_t', [(wq, 0, 0)], d, transpose=True, order=tm, numel=d * d)
                if not getattr(self, 'merge_ckv_bwd', False):      # (merged backward: 'dec.ca.kv_all_t0/1' below)
                    spack(key + '.kv_t', [(wk, 0, 0), (wv, 0, d)], 2 * d, transpose=True, numel=2 * d * d)
            else:
                spack(key + '.qkv', [(wq, 0, 0), (wk, d, 0), (wv, 2 * d, 0)], d, order=tm, numel=3 * d * d)
                spack(key + '.qkv_t', [(wq, 0, 0), (wk, 0, d), (wv, 0, 2 * d)], 3 * d, transpose=True, numel=3 * d * d)
            spack(key + '.o_t', [(wo, 0, 0)], d, transpose=True, order=tm, numel=d * d)
            spack(key + '.o', [(wo, 0, 0)], d, numel=d * d)       # LAST: the block's FFN stream follows it (hftt_attn_out_ffn_fwd reads the two as one)

        def strip_ffn(pre, key):
            if self.strip_small:
                return strip_ffn_s(pre, key)
            w1, w2 = pre + 'fc_1.weight', pre + 'fc_2.weight'          # [p, d], [d, p]
            base = spack(key + '.ffn', [(w1, 0, 0)], d, order=1, stride=2, offset=0, numel=2 * d * p)
            spack(key + '.ffn', [(w2, 0, 0)], p, order=0, stride=2, offset=1, base=base)
            # dX half of the backward: first matrix fc_2.weight^T [p, d], second fc_1.weight^T [d, p]
            base = spack(key + '.ffn_t', [(w2, 0, 0)], d, transpose=True, order=1, stride=2, offset=0, numel=2 * d * p)
            spack(key + '.ffn_t', [(w1, 0, 0)], p, transpose=True, order=0, stride=2, offset=1, base=base)

        self.merge_ckv = bool(self.strip and x3 and not self.strip_small and self.merge_ckv_opt and self.Ld in (2, 3) and self._planes(self.Hd, self.N, self.F))
        # ... and their backward (three layers): one weight-gradient product with six segments, the encoder-output gradient as two K = 768 halves
        self.merge_ckv_bwd = self.merge_ckv and self.Ld == 3 and os.environ.get('HFTT_X3_MERGE_CKV_BWD', '1') != '0'
        W['embed'] = wl.add('embed', _align(d, 64) * self.Kp, 64)
        W['embed_b'] = fl.add('embed_b', d, 8)
        blocks = []                                  # (prefix, key, has self attention, has cross attention)
        for i in range(self.Le):
            blocks.append((f'encoder_spec2midi.layers_freq.{i}.', f'enc{i}', True, False))
        blocks.append(('decoder_spec2midi.layer_zero_freq.', 'dec0', False, True))
        for i in range(self.Ld - 1):
            blocks.append((f'decoder_spec2midi.layers_freq.{i}.', f'dec{i + 1}', True, True))
        for i in range(self.Ld):
            blocks.append((f'decoder_spec2midi.layers_time.{i}.', f'time{i}', True, False))
        for pre, key, has_self, has_cross in blocks:
            if key == 'time0':
                heads('freq', 'heads_f')
            if has_self:
                attn_self(pre + 'self_attention.', key + '.sa')
            if has_cross:
                attn_cross(pre + 'encoder_attention.', key + '.ca')
            ffn(pre + 'positionwise_feedforward.', key)
            if self.strip:
                if has_self:
                    strip_attn(pre + 'self_attention.', key + '.sa', False)
                if has_cross:
                    strip_attn(pre + 'encoder_attention.', key + '.ca', True)
                strip_ffn(pre + 'positionwise_feedforward.', key)
        heads('time', 'heads_t')
        if self.merge_ckv:
            cross = [('decoder_spec2midi.layer_zero_freq.' if j == 0 else f'decoder_spec2midi.layers_freq.{j - 1}.') + 'encoder_attention.' for j in range(self.Ld)]
            parts = []
            for j, pre in enumerate(cross):
                parts += [(pre + 'fc_k.weight', 2 * j * d, 0), (pre + 'fc_v.weight', (2 * j + 1) * d, 0)]
            spack('dec.ca.kv_all', parts, d, order=1, numel=self.Ld * 2 * d * d)
            if self.merge_ckv_bwd:                       # backward: dX of the stacked projection as two K = 768 halves of the [d, 6d] transposed matrix
                names = [pre + n + '.weight' for pre in cross for n in ('fc_k', 'fc_v')]
                for half in range(2):
                    spack('dec.ca.kv_all_t%d' % half, [(names[3 * half + i], 0, i * d) for i in range(3)], 3 * d, transpose=True, numel=3 * d * d)
            W['dec.ca.kv_all_b'] = vec('dec.ca.kv_all_b', [pre + n + '.bias' for pre in cross for n in ('fc_k', 'fc_v')], d)
        if self.bfs:                                 # bf16 copy of the note position table: the (broadcast) residual of decoder layer zero
            off = wl.add('dec_pos_bf', self.N * d, 64)
            W['dec_pos_bf'] = off
            entries.append((self.poff['decoder_spec2midi.pos_embedding_freq.weight'], off, self.N, d, d, d, 0))

        self.Woff = W
        # prepared matrices: bf16 plane (npass 1) or fp32 copy (npass 3, parity) -- same element offsets
        n_w = _align(wl.off, 64)
        self.wbf = torch.zeros(n_w if self.npass == 1 else 8, dtype=torch.int16, device=self.device)
        self.wf32 = torch.zeros(n_w if self.npass == 3 else 8, dtype=torch.float32, device=self.device)
        # x3: every prepared matrix as two 16-bit planes (hi, lo) at the same element offsets -- fp16 halves for the forward matrices,
        # bf16 halves for the transposed ones (they only meet gradients)
        self.whi = torch.zeros(n_w if self.npass == 2 else 8, dtype=torch.int16, device=self.device)
        self.wlo = torch.zeros(n_w if self.npass == 2 else 8, dtype=torch.int16, device=self.device)
        self.fprep = torch.zeros(_align(fl.off, 8), dtype=torch.float32, device=self.device)
        self._prep_entries = list(entries)
        self._wl_regions = sorted((off, name) for name, off in wl.items.items())      # matrix planes by element offset
        self._wp_used, self._prep_built_for = set(), -1
        self._set_prep_table(entries)
        self.n_spack, self.n_spack_t = len(sentries), len(sentries_t)
        if sentries:
            self.wstrip = torch.zeros(_align(sl.off, 512), dtype=torch.int16, device=self.device)
            arr2 = (StripPackEntry * len(sentries))(*[StripPackEntry(*e) for e in sentries])
            self.spack_table = torch.frombuffer(bytearray(bytes(arr2)), dtype=torch.uint8).to(self.device)
        if sentries_t:
            arr3 = (StripPackEntry * len(sentries_t))(*[StripPackEntry(*e) for e in sentries_t])
            self.spack_table_t = torch.frombuffer(bytearray(bytes(arr3)), dtype=torch.uint8).to(self.device)
        # embed-fold scratch (dWeff, dbeff)
        self.dweff = torch.zeros(self.d * self.Kp, dtype=torch.float32, device=self.device)
        self.dbeff = torch.zeros(self.d, dtype=torch.float32, device=self.device)
        e = 'encoder_spec2midi.'
        self.fold = FoldDesc(self.d, self.cfg['cnn_channel'], self.cfg['cnn_kernel'], self.n_proc, self.Kp, _align(self.d, 64),
                             self.P(e + 'conv.weight'), self.P(e + 'conv.bias'), self.P(e + 'tok_embedding_freq.weight'),
                             self.P(e + 'tok_embedding_freq.bias'),
                             (self.wbf.data_ptr() + 2 * W['embed']) if self.npass == 1 else 0,
                             (self.wf32.data_ptr() + 4 * W['embed']) if self.npass == 3 else 0,
                             self.fprep.data_ptr() + 4 * W['embed_b'],
                             self.dweff.data_ptr(), self.dbeff.data_ptr(),
                             self.G(e + 'conv.weight'), self.G(e + 'conv.bias'), self.G(e + 'tok_embedding_freq.weight'),
                             self.G(e + 'tok_embedding_freq.bias'),
                             (self.whi.data_ptr() + 2 * W['embed']) if self.npass == 2 else 0,
                             (self.wlo.data_ptr() + 2 * W['embed']) if self.npass == 2 else 0)

    def _set_prep_table(self, entries):
        arr = (PrepEntry * len(entries))()
        for i, (so, do, r, c_, sld, dld, kind) in enumerate(entries):
            arr[i] = PrepEntry(so, do, r, c_, sld, dld, kind, 4 if kind == 1 else 2)      # pad: x3 element type (transposed = backward = bf16 halves)
        self.prep_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self.n_prep = len(entries)

    # per-block matrix planes that only the general GEMM reads: with the strip plans most of them are never asked for (the strip packs
    # replace them), and preparing all of them every step cost 80 us.  A plane is prepared once any launch plan has asked for it (Wp).
    _BLOCK_PLANES = ('.qkv', '.qkv_t', '.o', '.o_t', '.q', '.q_t', '.kv', '.kv_t', '.f1', '.f1_t', '.f2', '.f2_t')

    def _refresh_prep_table(self):
        if not self.strip or self._prep_built_for == len(self._wp_used):
            return
        import bisect
        offs = [o for o, _ in self._wl_regions]
        keep = []
        for ent in self._prep_entries:
            do, kind = ent[1], ent[6]
            if kind != 2:                            # (kind 2: vectors in the fp32 plane, always prepared)
                name = self._wl_regions[bisect.bisect_right(offs, do) - 1][1]
                if name.endswith(self._BLOCK_PLANES) and name not in self._wp_used:
                    continue
            keep.append(ent)
        self._set_prep_table(keep)
        self._prep_built_for = len(self._wp_used)

    def Wp(self, key):   # device address of a prepared matrix (bf16 plane or fp32 copy, by precision)
        o = self.Woff[key]
        self._wp_used.add(key)
        if self.npass == 2:
            return self.whi.data_ptr() + 2 * o
        return (self.wbf.data_ptr() + 2 * o) if self.npass == 1 else (self.wf32.data_ptr() + 4 * o)

    def Wp_lo(self, W):   # x3: the lo plane of the prepared matrix whose hi plane sits at device address W
        return (W - self.whi.data_ptr() + self.wlo.data_ptr()) if self.npass == 2 else 0

    def Fp(self, key):
        return self.fprep.data_ptr() + 4 * self.Woff[key]

    def Ws(self, key):   # device address of a strip-packed weight stream
        return self.wstrip.data_ptr() + 2 * self.Woff['s.' + key]

    def prepare_weights(self, stream):
        """parameters -> the kernels' operand forms (bf16 / fp32 planes, folded embedding, strip packs): every forward, because the flat buffer
        may have changed in ways nobody tells the engine about (an optimizer step, ``p.data.copy_``) -- unless the caller has declared the
        parameters constant (``frozen_weights``: inference servers; model.amt.AMT does), then once."""
        if self.frozen_weights and self._prepared_frozen:
            return
        self._refresh_prep_table()
        if self.npass == 2:
            check(self.lib.hftt_prep_weights_x3(self.flat_params.data_ptr(), self.whi.data_ptr(), self.wlo.data_ptr(),
                                                self.fprep.data_ptr(), self.prep_table.data_ptr(), self.n_prep, stream), 'prep_weights_x3')
        else:
            check(self.lib.hftt_prep_weights(self.flat_params.data_ptr(), self.wbf.data_ptr() if self.npass == 1 else 0,
                                             self.wf32.data_ptr() if self.npass == 3 else 0,
                                             self.fprep.data_ptr(), self.prep_table.data_ptr(), self.n_prep, stream), 'prep_weights')
        check(self.lib.hftt_embed_fold_fwd(C.byref(self.fold), stream), 'embed_fold_fwd')
        if self.strip and self.n_spack:
            if self.x3:
                check(self.lib.hftt_x3_strip_pack(self.flat_params.data_ptr(), self.wstrip.data_ptr(), self.spack_table.data_ptr(), self.n_spack, 2, stream), 'x3_strip_pack')
                check(self.lib.hftt_x3_strip_pack(self.flat_params.data_ptr(), self.wstrip.data_ptr(), self.spack_table_t.data_ptr(), self.n_spack_t, 4, stream), 'x3_strip_pack')
            elif self.strip_small:                   # bf16 mode, small widths: compact packs with bf16 halves for both directions
                check(self.lib.hftt_x3_strip_pack(self.flat_params.data_ptr(), self.wstrip.data_ptr(), self.spack_table.data_ptr(), self.n_spack, 4, stream), 'x3_strip_pack')
                if self.n_spack_t:
                    check(self.lib.hftt_x3_strip_pack(self.flat_params.data_ptr(), self.wstrip.data_ptr(), self.spack_table_t.data_ptr(), self.n_spack_t, 4, stream), 'x3_strip_pack')
            else:
                check(self.lib.hftt_strip_pack(self.flat_params.data_ptr(), self.wstrip.data_ptr(), self.spack_table.data_ptr(), self.n_spack, stream), 'strip_pack')
        self._prepared_frozen = self.frozen_weights

    # ------------------------------------------------------------------ plan building helpers
    def _new_site(self):
        self._site += 1
        return self._site

    def _buf(self, ws, name, *shape, dtype=torch.float32, half=False, hidden=False):
        """half=True: a GEMM-only tensor -> bf16 when the engine stores such tensors as bf16; hidden=True: the FFN hidden / its gradient
        (bf16 in the x3 strip plans too)."""
        if (half and self.sb) or (hidden and self.hh):
            dtype = torch.bfloat16
        t = ws['bufs'].get(name)
        if t is not None:
            if tuple(t.shape) != tuple(shape) or t.dtype != dtype:
                raise _capi.HfttError('workspace buffer %s re-declared with a different shape / dtype' % name)
            return t
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        ws['bufs'][name] = t
        return t

    def _abuf(self, ws, name, *shape):
        """activation-stream tensor: bf16 when the strip kernels run (bf16 residual stream), fp32 otherwise"""
        return self._buf(ws, name, *shape, dtype=torch.bfloat16 if self.bfs else torch.float32)

    def _pbuf(self, ws, name, *shape):
        """saved pre-LayerNorm sum (read by the LayerNorm backward only): bf16 on the bf16 stream and in the x3 strip plans"""
        return self._buf(ws, name, *shape, dtype=torch.bfloat16 if (self.bfs or self.hh) else torch.float32)

    def _nt(self, plan, ws, M, N, K, A, lda, W, bias, Cp, ldc, act=0, out_scale=1.0, add_table=0, add_mod=0,
            gate=0, ldg=0, gate_scale=1.0, drop_site=0, residual=0, ldr=0, res_mod=0, ln=None, a_bf=False, c_bf=False, gate_bf=False, res_bf=False):
        a_bf, c_bf, gate_bf, res_bf = (a_bf and self.sb), (c_bf and self.sb), (gate_bf and self.sb), (res_bf and self.sb and bool(residual))
        dsc = GemmNtDesc()
        npass = self.npass if self.npass != 2 else (4 if self._in_backward else 2)       # x3: fp16 halves forward, bf16 halves with gradients
        a_hi = npass == 4 and self.g8                 # (every block GEMM of the backward has a gradient as its A operand)
        dsc.io_flags = (1 if a_bf else 0) | (2 if c_bf else 0) | (4 if gate_bf else 0) | (8 if res_bf else 0) | (16 if a_hi else 0)
        dsc.M, dsc.N, dsc.K, dsc.npass = M, N, K, npass
        dsc.A, dsc.lda = A, lda
        dsc.W = W
        dsc.W_lo = self.Wp_lo(W)
        dsc.bias = bias
        dsc.C, dsc.ldc = Cp, ldc
        dsc.act, dsc.out_scale = act, out_scale
        dsc.add_table, dsc.add_mod = add_table, add_mod
        if isinstance(gate_scale, tuple):      # run-time value: 1/(1-p) of the hidden-layer dropout
            ws.setdefault('gate_descs', []).append(dsc)
            gate_scale = 1.0
        dsc.gate, dsc.ldg, dsc.gate_scale = gate, ldg, gate_scale
        dsc.drop_p, dsc.drop_site, dsc.drop_seed = 0.0, drop_site, 0
        dsc.residual, dsc.ldr, dsc.res_mod = residual, ldr, (res_mod or M)
        if ln is not None:
            dsc.ln_gamma, dsc.ln_beta, dsc.pre_ln_out, dsc.ln_mean, dsc.ln_rstd = ln
        if drop_site:
            ws['drop'].append(dsc)
        ws['keep'].append(dsc)
        n_pad = _align(N, 64)
        bn = N if ln is not None else (256 if n_pad % 256 == 0 else (128 if n_pad % 128 == 0 else 64))
        esz = 2 if self.npass == 1 else 4           # (x3: two 16-bit planes = 4 bytes per weight)
        nbytes = (2 if a_bf else 4) * M * K + (2 if c_bf else 4) * M * N + esz * N * K + ((2 if res_bf else 4) * M * N if residual else 0) + (4 * M * N if ln is not None else 0) \
            + ((2 if gate_bf else 4) * M * N if gate else 0)
        rich = bool(add_table or gate or drop_site or residual or ln is not None)
        if self.npass == 1 and N % 256 == 0 and K <= 768 and M >= 256:      # mirrors dispatch_nt_bf16 in csrc/gemm_nt.hip
            pf = 2 if (K // 32) % 2 == 0 else 1
            if K <= 256:
                elementwise = not (add_table or residual or ln is not None) and c_bf and (not gate or gate_bf)
                if not rich:
                    kname = 'gemm_nt_as1_kernel<64, 0, false, %d>' % pf
                elif elementwise:
                    kname = 'gemm_nt_as1_kernel<64, 0, true, %d>' % pf
                elif N == 256:
                    kname = 'gemm_nt_as1_kernel<64, 1, false, %d>' % pf
                else:
                    kname = 'gemm_nt_as1_kernel<32, 2, false, 1>'
            elif N == 256 and pf == 2:
                kname = 'gemm_nt_as1_kernel<%d, 1, false, 2>' % (32 if (ln is not None and K <= 512) else 64)
            elif K <= 512:
                kname = 'gemm_nt_as_kernel<4, false, true>' if a_bf else 'gemm_nt_as_kernel<8, false, false>'
            else:
                kname = 'gemm_nt_as_kernel<6, true, true>' if a_bf else 'gemm_nt_as_kernel<12, true, false>'
        else:
            kname = 'gemm_nt_kernel<%d, %d, %s>' % (bn, 5 if a_hi else npass, 'true' if ln is not None else 'false')
        meta = {'kernel': kname, 'flops': 2.0 * M * N * K, 'bytes': float(nbytes), 'shape': (M, N, K)}
        plan.append((self.lib.hftt_gemm_nt, (C.byref(dsc),), 'gemm_nt', meta))
        return dsc

    def _sl(self, plan, ws, M, N, K, x, ldx, wkey, bias, Cp, ldc, relu=False, out_scale=1.0, gate=0, ldg=0, gate_scale=1.0, drop_site=0,
            residual=0, ldr=0, res_mod=0, res_bf=True, ln=None, x_bf=True, c_bf=True, c_planes=False, x_drop_site=0):
        """hftt_strip_linear plan entry (bf16 mode, N % 256 == 0): C = epi(x . Wl^T + bias), Wl = strip pack `wkey`."""
        dsc = StripDesc()
        dsc.M, dsc.N, dsc.K = M, N, K
        if self.x3:                                  # fp32 tensors, fp16 halves on forward products, bf16 halves where a gradient is an operand
            x_bf = c_bf = res_bf = False
            dsc.flags = (SL_X3_BF16 if self._in_backward else SL_X3_F16) | (SL_RELU if relu else 0) | (SL_PRE_BF16 if (self.hh and ln is not None) else 0) \
                | (SL_X3_GRAD_HI if (self._in_backward and self.g8) else 0) | (SL_C_F16PAIR if c_planes else 0) | (SL_X_DROP if x_drop_site else 0)
            if x_drop_site:                          # HFTT_SL_X_DROP: x (the LayerNorm backward's dr) is masked while it is loaded
                assert self._in_backward and not drop_site and N == 256 and K == 256
                drop_site = x_drop_site
        else:
            dsc.flags = (SL_X_BF16 if x_bf else 0) | (SL_C_BF16 if c_bf else 0) | (SL_RES_BF16 if (residual and res_bf) else 0) | (SL_RELU if relu else 0)
        dsc.x, dsc.ldx, dsc.w, dsc.bias = x, ldx, self.Ws(wkey), bias
        dsc.C, dsc.ldc, dsc.out_scale = Cp, ldc, out_scale
        if isinstance(gate_scale, tuple):
            ws.setdefault('gate_descs', []).append(dsc)
            gate_scale = 1.0
        dsc.gate, dsc.ldg, dsc.gate_scale = gate, ldg, gate_scale
        dsc.drop_p, dsc.drop_site, dsc.drop_seed = 0.0, drop_site, 0
        dsc.residual, dsc.ldr, dsc.res_mod = residual, ldr, res_mod
        pre_saved = False
        if ln is not None:
            dsc.ln_gamma, dsc.ln_beta, dsc.pre_ln_out, dsc.ln_mean, dsc.ln_rstd = ln
            pre_saved = bool(ln[2])
        if drop_site:
            ws['drop'].append(dsc)
        ws['keep'].append(dsc)
        tf = lambda v: 'true' if v else 'false'
        nbytes = (2 if x_bf else 4) * M * K + (2 if c_bf else 4) * M * N + ((2 if (c_bf or self.hh) else 4) * M * N if pre_saved else 0) + (4 if self.x3 else 2) * N * K \
            + ((2 if res_bf else 4) * M * N if residual else 0) + (2 * M * N if gate else 0)
        # kernel symbol as rocprofv3 prints it (the C side picks the pipelined form by the rule mirrored here: strip_gemm2.hip hftt_strip_linear2_try)
        passes, kch = N // 256, K // 256
        v2 = (os.environ.get('HFTT_STRIP_V2', '1')[:1] != '0' and x_bf and c_bf and K % 256 == 0 and M % 32 == 0 and not gate and (not residual or res_bf)
              and ((ln is not None and kch <= 3) or (ln is None and (kch, passes) in ((1, 1), (1, 2), (1, 3), (2, 1), (3, 1)))))
        kname = ('strip_linear2_kernel<%s, %d, %d, %s, %s>' % (tf(ln is not None), 1 if ln is not None else passes, kch, tf(bool(residual)),
                                                                tf(os.environ.get('HFTT_LINEAR2_PATCH', '1')[:1] != '0'))) if v2 \
            else 'strip_linear_kernel<%s, %s, %s>' % (tf(x_bf), tf(c_bf), tf(ln is not None))
        if self.x3 and self.strip_small:
            kname = 'x3s_linear_kernel<%d, %d, %d, %s, %s>' % (4 if self._in_backward else 2, K // 32, N // 32, tf(ln is not None), tf(bool(residual)))
        elif self.strip_small:
            kname = 'bs_linear_kernel<%d, %d, %s, %s>' % (K // 32, N // 32, tf(ln is not None), tf(bool(residual)))
        elif self.x3:
            xe = (5 if self.g8 else 4) if self._in_backward else 2
            # (last argument: resident strip chunks -- the one-pass forms without LayerNorm keep half a set and run two workgroups per CU: x3_strip.hip launch_xl)
            kname = 'x3_linear_kernel<%d, %s, %d, %d, %s, %d>' % (xe, tf(ln is not None), passes, kch, tf(bool(residual)), 8 if (ln is None and passes == 1) else 16)
            if ln is None and kch == 1:
                kname = 'x3_linear_n_kernel<%d, %d, %s, %s, %s>' % (xe, N // 32, tf(bool(residual)), tf(c_planes), tf(bool(x_drop_site)))
        meta = {'kernel': kname, 'flops': 2.0 * M * N * K, 'bytes': float(nbytes), 'shape': (M, N, K)}
        plan.append((self.lib.hftt_strip_linear, (C.byref(dsc),), 'strip_linear', meta))
        return dsc

    def _mlp(self, plan, ws, mode, M, x, wkey, y, b1=0, b2=0, h_out=0, gate=0, gate_scale=1.0, site_h=0, site_o=0, residual=0, ln=None):
        """fused two-GEMM block: mode 0 = hftt_ffn_res_ln_fwd (x -> relu/dropout hidden -> + x -> LayerNorm), mode 1 = hftt_ffn_bwd_dx."""
        d, p = self.d, self.p
        dsc = FfnDesc()
        dsc.M, dsc.d, dsc.p, dsc.mode = M, d, p, mode
        dsc.flags = ((SL_X3_F16 if mode == 0 else SL_X3_BF16) | (SL_H_BF16 | SL_PRE_BF16 if self.hh else 0) | (SL_X3_GRAD_HI if (mode == 1 and self.g8) else 0)) if self.x3 else (SL_X_BF16 | SL_C_BF16 | SL_RES_BF16)
        dsc.x, dsc.ldx, dsc.w = x, d, self.Ws(wkey)
        dsc.b1, dsc.b2 = b1, b2
        dsc.h_out, dsc.ldh = h_out, p
        dsc.gate, dsc.ldg = gate, p
        if isinstance(gate_scale, tuple):
            ws.setdefault('gate_descs', []).append(dsc)
            gate_scale = 1.0
        dsc.gate_scale = gate_scale
        dsc.drop_p, dsc.site_h, dsc.site_o, dsc.drop_seed = 0.0, site_h, site_o, 0
        dsc.residual, dsc.ldr = residual, d
        pre_saved = False
        if ln is not None:
            dsc.ln_gamma, dsc.ln_beta, dsc.pre_ln_out, dsc.ln_mean, dsc.ln_rstd = ln
            pre_saved = bool(ln[2])
        dsc.y, dsc.ldy = y, d
        if (mode == 0 and (site_h or site_o)) or (mode == 1 and site_o):      # mode 1: site_o = the dropout whose OUTPUT gradient the strip dy is (masked on load)
            ws['drop'].append(dsc)
        ws['keep'].append(dsc)
        esz = 4.0 if self.x3 else 2.0
        hsz = 2.0 if (self.hh or not self.x3) else 4.0
        nbytes = esz * M * d * (2 + (1 if residual else 0)) + (hsz * M * d if pre_saved else 0) + (hsz * M * p if h_out else 0) + (hsz * M * p if gate else 0) + 2 * esz * d * p
        v2 = os.environ.get('HFTT_STRIP_V2', '1')[:1] != '0' and p == 512 and M % 32 == 0 and not (mode == 0 and residual)
        xname = ('x3s_mlp_kernel<%%d, %s>' % ('true' if self.hh else 'false')) if self.strip_small else \
            ('x3_mlp_kernel<%%d, 16, %s, %s>' % ('true' if self.hh else 'false', 'true' if (mode == 1 and self.g8) else 'false'))
        # (kernel symbols as rocprofv3 prints them -- tests/test_kernel_names_gpu.py holds every plan-meta name against a kernel trace.  The bf16 fused
        # block takes its whole-line store path, the last template argument, for the training forward only: strip_gemm2.hip hftt_strip_mlp2_try)
        stp = os.environ.get('HFTT_MLP2_PATCH')
        stp = (stp[:1] != '0') if stp else (mode == 0 and bool(h_out or pre_saved))
        bname = 'bs_mlp_kernel<%d>' if self.strip_small else (('strip_mlp2_kernel<%%d, 16, %s>' % ('true' if stp else 'false')) if v2 else 'strip_mlp_kernel<%d>')
        meta = {'kernel': (xname if self.x3 else bname) % mode, 'flops': 4.0 * M * d * p, 'bytes': nbytes,
                'shape': (M, d, p), 'saves': bool(h_out or pre_saved)}
        plan.append((self.lib.hftt_ffn_res_ln_fwd if mode == 0 else self.lib.hftt_ffn_bwd_dx, (C.byref(dsc),), 'ffn_fwd' if mode == 0 else 'ffn_bwd_dx', meta))
        return dsc

    def _tn(self, plan, ws, M, N, K, dY, lddy, X, ldx, segs, K_out=None, out_scale=1.0, beta=0.0, dy_bf=False, x_bf=False, dy_hid=False, x_hid=False, dy_drop_site=0):
        """segs: list of (row0, rows, dw_addr, db_addr or 0)"""
        need = self.lib.hftt_gemm_tn_ws_bytes(M, N, K)
        ws['tn_need'] = max(ws.get('tn_need', 0), need)
        dsc = GemmTnDesc()
        dsc.M, dsc.N, dsc.K, dsc.npass = M, N, K, (4 if self.npass == 2 else self.npass)
        # dy_hid / x_hid: this operand is the FFN hidden's gradient / the stored hidden (bf16 in the x3 strip plans as well)
        dy_bf, x_bf = (dy_bf and self.sb) or (dy_hid and self.hh), (x_bf and self.sb) or (x_hid and self.hh)
        dsc.io_flags = (1 if dy_bf else 0) | (2 if x_bf else 0) | (4 if (self.g8 and not dy_bf) else 0) | (8 if dy_drop_site else 0)
        if dy_drop_site:                             # HFTT_TN_DY_DROP: dY (fp32, the LayerNorm backward's dr) is masked while it is loaded
            assert not dy_bf and lddy == N and not self.g8
            dsc.drop_p, dsc.drop_site, dsc.drop_seed = 0.0, dy_drop_site, 0
            ws['drop'].append(dsc)
        dsc.dY, dsc.lddy, dsc.X, dsc.ldx = dY, lddy, X, ldx
        dsc.out_scale, dsc.beta = out_scale, beta
        dsc.n_seg = len(segs)
        for i, (r0, rows, dw, db) in enumerate(segs):
            dsc.seg_row0[i], dsc.seg_rows[i], dsc.seg_dw[i], dsc.seg_db[i] = r0, rows, dw, db
        dsc.K_out = K_out or K
        ws['tn'].append(dsc)
        ws['keep'].append(dsc)
        tile = '2, 4' if (N >= 256 and K >= 256) else ('1, 2' if (N >= 128 and K >= 128) else '1, 1')
        if tile == '2, 4' and N <= 256 and K <= 256:
            tile = '1, 4'                           # (csrc/gemm_tn.hip tn_plan: the 128 x 256 tile for single-tile shapes)
        meta = {'kernel': 'gemm_tn_kernel<%s, %d, %s, %s>' % (tile, 6 if dy_drop_site else (5 if (dsc.io_flags & 4) else dsc.npass), 'true' if dy_bf else 'false', 'true' if x_bf else 'false'), 'flops': 2.0 * M * N * K,
                'bytes': (2.0 if dy_bf else 4.0) * M * N + (2.0 if x_bf else 4.0) * M * K + 4.0 * N * K, 'shape': (M, N, K)}
        plan.append((self.lib.hftt_gemm_tn, (C.byref(dsc),), 'gemm_tn', meta))
        return dsc

    def _attn(self, plan, ws, bwd, n_seq, H, Lq, Lk, q, qss, ldq, k, kss, ldk, v, vss, ldv, out, oss, ldo, lse, probs=0,
              drop_site=0, dout=0, dq=0, dqss=0, lddq=0, dk=0, dkss=0, lddk=0, dv=0, dvss=0, lddv=0, flags=0, planes=False, map_out=False):
        flags = (flags if self.sb else 0) | ((ATTN_Q_F16PAIR | ATTN_KV_F16PAIR) if planes else 0)
        dsc = AttnDesc()
        dsc.io_flags = flags
        dsc.n_seq, dsc.n_heads, dsc.Lq, dsc.Lk, dsc.dh, dsc.npass = n_seq, H, Lq, Lk, self.d // H, self.npass
        dsc.q, dsc.q_seq_stride, dsc.ldq = q, qss, ldq
        dsc.k, dsc.k_seq_stride, dsc.ldk = k, kss, ldk
        dsc.v, dsc.v_seq_stride, dsc.ldv = v, vss, ldv
        dsc.out, dsc.o_seq_stride, dsc.ldo = out, oss, ldo
        dsc.lse, dsc.probs = lse, probs
        dsc.drop_p, dsc.drop_site, dsc.drop_seed = 0.0, drop_site, 0
        dsc.dout = dout
        dsc.dq, dsc.dq_seq_stride, dsc.lddq = dq, dqss, lddq
        dsc.dk, dsc.dk_seq_stride, dsc.lddk = dk, dkss, lddk
        dsc.dv, dsc.dv_seq_stride, dsc.lddv = dv, dvss, lddv
        if drop_site:
            ws['drop'].append(dsc)
        ws['keep'].append(dsc)
        dh = self.d // H
        kt = (Lk + 31) // 32
        kt = kt if kt <= 4 else 8
        # dropout form of the x3 kernels (a template parameter, chosen by the C side from drop_p and the shape): 0 none, 1 per key quad, 2 per element
        dm = 0 if (not (drop_site and self.dropout > 0.0) or getattr(self, '_building_inference', False)) else (1 if (Lk % 4 == 0 and (n_seq * H * Lq * Lk) >> 34 == 0) else 2)
        eq = 2.0 if flags & 1 else 4.0
        ekv = 2.0 if flags & 2 else 4.0
        eo = 2.0 if flags & 4 else 4.0
        qkv_bytes = n_seq * (eq * Lq + 2 * ekv * Lk) * self.d
        if bwd:
            hb = 'true' if (flags & 7) == 7 else 'false'
            meta = {'kernel': ('x3_attn_bwd_kernel<%d, %d, %s, %d>' % (kt, dh, 'true' if planes else 'false', dm)) if self.npass == 2 else 'attn_bwd_kernel<%d, %d, %d, %s>' % (kt, dh, self.npass, hb), 'flops': 10.0 * n_seq * H * Lq * Lk * dh,
                    'bytes': qkv_bytes + n_seq * ((2.0 if flags & 8 else 4.0) * Lq + 2 * (2.0 if flags & 16 else 4.0) * Lk) * self.d + 2 * eo * n_seq * Lq * self.d
                    + 8.0 * n_seq * H * Lq, 'shape': (n_seq, H, Lq, Lk, dh)}
        else:
            hb = 'true' if (flags & 7) == 7 else 'false'
            long_rows = (hb == 'true' and dh == 64 and self.npass == 1 and 128 < Lk <= 256 and 128 < Lq <= 256 and not probs
                         and os.environ.get('HFTT_ATTN_FWD8', '1')[:1] != '0')                   # csrc/attn_fwd8.hip: hftt_attn_fwd8_try
            x3name = ('x3p_attn_fwd_kernel<%d, %d, %s, %d>' % (kt, 8 if (kt == 8 and Lq > 128) else 4, 'true' if (probs or map_out) else 'false', dm)) if planes else \
                ('x3_attn_fwd_kernel<%d, %d, %d, %s>' % (kt, dh, 8 if kt == 8 else 4, 'true' if (probs or map_out) else 'false'))
            meta = {'kernel': 'attn_fwd8_kernel' if long_rows else (x3name if self.npass == 2 else
                                                                     'attn_fwd_kernel<%d, %d, %d, %s>' % (kt, dh, self.npass, hb)), 'flops': 4.0 * n_seq * H * Lq * Lk * dh,
                    # q, k, v, out + the row statistics (max, 1/sum) + the attention map where it is a model output (fp32, mandatory)
                    'bytes': qkv_bytes + eo * n_seq * Lq * self.d + 8.0 * n_seq * H * Lq + (4.0 * n_seq * H * Lq * Lk if (probs or map_out) else 0.0),
                    'shape': (n_seq, H, Lq, Lk, dh)}
        plan.append((self.lib.hftt_attn_bwd if bwd else self.lib.hftt_attn_fwd, (C.byref(dsc),), 'attn_bwd' if bwd else 'attn_fwd', meta))
        return dsc

    def _planes(self, H, Lq, Lk):
        """x3 strip plans: do the q / k / v projections of this attention hand their results over as f16-pair planes (written once by the
        projection's epilogue, staged by LDS-DMA in the attention forward: csrc/x3_attn_pl.hip)?  dh == 64 and one query block per wave."""
        if not (self.x3 and self.strip and self.planes_opt and self.d // H == 64):
            return False
        nqb = (Lq + 31) // 32
        return nqb <= (8 if Lk > 128 else 4)

    def _mic(self):
        """masked-in-consumers: this plan's LayerNorm backward writes no masked copy (x3 strip plans at d = 256, dropout on)"""
        return bool(self.strip and self.x3 and not self.strip_small and not self.g8 and self.ln_mask_in_consumers_opt and self.dropout > 0.0)

    def _lnb(self, plan, ws, M, dy, r, mean, rstd, gamma, dr, dr_drop, drop_site, dgamma, dbeta, beta, drop_bf=True, dy_bf=False, dr_bf=False):
        r_bf = self.bfs or self.hh                  # the strip forward kernels (bf16, and x3) save the pre-LayerNorm sum as bf16
        n_wg = self.lib.hftt_ln_bwd_wgs(M)
        ws['ln_need'] = max(ws.get('ln_need', 0), n_wg * 2 * self.d * 4)
        dsc = LnBwdDesc()
        dsc.M, dsc.N = M, self.d
        dsc.dy, dsc.r, dsc.mean, dsc.rstd, dsc.gamma = dy, r, mean, rstd, gamma
        dsc.dr, dsc.dr_drop = dr, dr_drop
        dsc.drop_bf16 = 1 if (drop_bf and self.sb and dr_drop) else 0
        dsc.io_flags = (1 if (dy_bf and self.sb) else 0) | (2 if (dr_bf and self.sb) else 0) | (4 if r_bf else 0)
        dsc.drop_p, dsc.drop_site, dsc.drop_seed = 0.0, drop_site, 0
        ws['ln'].append(dsc)
        if drop_site:
            ws['drop'].append(dsc)
        ws['keep'].append(dsc)
        plan.append((self.lib.hftt_ln_bwd, (C.byref(dsc),), 'ln_bwd', None))
        plan.append(('ln_reduce', (n_wg, self.d, dgamma, dbeta, beta), 'ln_bwd_reduce', None))

    # ------------------------------------------------------------------ workspace + plans for one batch size
    def workspace(self, B):
        if B in self._ws:
            return self._ws[B]
        if not self.is_bound():
            raise _capi.HfttError('engine parameters are not bound')
        ws = {'bufs': {}, 'drop': [], 'keep': [], 'tn': [], 'ln': [], 'B': B}
        # The x3 strip kernels take whole 32-token strips (hftt_x3_strip_linear / hftt_x3_strip_mlp: M % 32 == 0).  A batch whose token counts
        # are not multiples of 32 (B * T % 4 != 0 with 88 notes: odd batch x odd frame count) gets the block-GEMM plans of the same precision
        # for THIS workspace only -- same arithmetic (npass 2 / 4 in gemm_nt / gemm_tn), fp32 saved tensors.
        Se, Sn = B * self.T * self.F, B * self.T * self.N
        strip_here = self.strip and not (self.x3 and (Se % 32 or Sn % 32))
        saved_mode = (self.strip, self.bfs, self.hh)
        if not strip_here:
            self.strip = self.bfs = self.hh = False
        ws['strip'] = strip_here
        try:
            self._site = 0
            self._build_forward(ws, save=True)
            if self.strip:                           # inference plan: same buffers, nothing saved for a backward
                n_sites = self._site
                self._site = 0
                self._building_inference = True        # (kernel symbols of the plan meta: the eval forward runs without dropout)
                try:
                    self._build_forward(ws, save=False)
                finally:
                    self._building_inference = False
                assert self._site == n_sites
            self._in_backward = True
            try:
                self._build_backward(ws)
            finally:
                self._in_backward = False
        finally:
            self.strip, self.bfs, self.hh = saved_mode
        tnb = torch.empty(max(ws.get('tn_need', 8), 8) // 4 + 16, dtype=torch.float32, device=self.device)
        lnb = torch.empty(max(ws.get('ln_need', 8), 8) // 4 + 16, dtype=torch.float32, device=self.device)
        ws['bufs']['tn_ws'], ws['bufs']['ln_ws'] = tnb, lnb
        for dsc in ws['tn']:
            dsc.ws, dsc.ws_bytes = tnb.data_ptr(), tnb.numel() * 4
        for dsc in ws['ln']:
            dsc.ws = lnb.data_ptr()
        ws['ln_ws_ptr'] = lnb.data_ptr()
        self._ws[B] = ws
        return ws

    def _enc_layer_fwd(self, plan, ws, tag, key, pre, S, n_seq, L, H, x_in, save=True):
        """EncoderLayer (model_spec2midi.py:230-245).  Returns address of the layer output [S, d]."""
        d, p = self.d, self.p
        b = ws['bufs']
        hz = 2 if self.sb else 4                    # element size of the GEMM-only ("half") tensors
        qkv = self._buf(ws, tag + '.qkv', S, 3 * d, half=True)
        ctx = self._buf(ws, tag + '.ctx', S, d, half=True)
        lse = self._buf(ws, tag + '.lse', n_seq * H * L * 2)
        r1 = self._pbuf(ws, tag + '.r1', S, d); x1 = self._abuf(ws, tag + '.x1', S, d)
        m1 = self._buf(ws, tag + '.m1', S); s1 = self._buf(ws, tag + '.s1', S)
        h = self._buf(ws, tag + '.h', S, p, half=True, hidden=True)
        r2 = self._pbuf(ws, tag + '.r2', S, d); x2 = self._abuf(ws, tag + '.x2', S, d)
        m2 = self._buf(ws, tag + '.m2', S); s2 = self._buf(ws, tag + '.s2', S)
        sites = ws.setdefault('sites', {})
        sa, so, sh, sf = (self._new_site() for _ in range(4))
        sites[tag] = (sa, so, sh, sf)
        gam, bet = self.P(pre + 'layer_norm.weight'), self.P(pre + 'layer_norm.bias')
        if self.strip:
            sv = (lambda t: t.data_ptr()) if save else (lambda t: 0)
            pln = self._planes(H, L, L)
            self._sl(plan, ws, S, 3 * d, d, x_in, d, key + '.sa.qkv', self.Fp(key + '.sa.qkv_b'), qkv.data_ptr(), 3 * d, c_planes=pln)
            q = qkv.data_ptr()
            self._attn(plan, ws, False, n_seq, H, L, L, q, L * 3 * d, 3 * d, q + hz * d, L * 3 * d, 3 * d, q + 2 * hz * d, L * 3 * d, 3 * d,
                       ctx.data_ptr(), L * d, d, lse.data_ptr(), drop_site=sa, flags=1 | 2 | 4, planes=pln)
            self._sl(plan, ws, S, d, d, ctx.data_ptr(), d, key + '.sa.o', self.P(pre + 'self_attention.fc_o.bias'), x1.data_ptr(), d,
                     drop_site=so, residual=x_in, ldr=d, ln=(gam, bet, sv(r1), sv(m1), sv(s1)))
            self._mlp(plan, ws, 0, S, x1.data_ptr(), key + '.ffn', x2.data_ptr(), b1=self.P(pre + 'positionwise_feedforward.fc_1.bias'),
                      b2=self.P(pre + 'positionwise_feedforward.fc_2.bias'), h_out=sv(h), site_h=sh, site_o=sf, ln=(gam, bet, sv(r2), sv(m2), sv(s2)))
            return x2.data_ptr()
        self._nt(plan, ws, S, 3 * d, d, x_in, d, self.Wp(key + '.sa.qkv'), self.Fp(key + '.sa.qkv_b'), qkv.data_ptr(), 3 * d, c_bf=True)
        q = qkv.data_ptr()
        self._attn(plan, ws, False, n_seq, H, L, L, q, L * 3 * d, 3 * d, q + hz * d, L * 3 * d, 3 * d, q + 2 * hz * d, L * 3 * d, 3 * d,
                   ctx.data_ptr(), L * d, d, lse.data_ptr(), drop_site=sa, flags=1 | 2 | 4)
        self._nt(plan, ws, S, d, d, ctx.data_ptr(), d, self.Wp(key + '.sa.o'), self.P(pre + 'self_attention.fc_o.bias'), x1.data_ptr(), d,
                 drop_site=so, residual=x_in, ldr=d, ln=(gam, bet, r1.data_ptr(), m1.data_ptr(), s1.data_ptr()), a_bf=True)
        self._nt(plan, ws, S, p, d, x1.data_ptr(), d, self.Wp(key + '.f1'), self.P(pre + 'positionwise_feedforward.fc_1.bias'), h.data_ptr(), p,
                 act=1, drop_site=sh, c_bf=True)
        self._nt(plan, ws, S, d, p, h.data_ptr(), p, self.Wp(key + '.f2'), self.P(pre + 'positionwise_feedforward.fc_2.bias'), x2.data_ptr(), d,
                 drop_site=sf, residual=x1.data_ptr(), ldr=d, ln=(gam, bet, r2.data_ptr(), m2.data_ptr(), s2.data_ptr()), a_bf=True)
        return x2.data_ptr()

    def _ffn_fwd(self, plan, ws, tag, key, pre, S, x_in, sites, save=True):
        d, p = self.d, self.p
        h = self._buf(ws, tag + '.h', S, p, half=True, hidden=True)
        r = self._pbuf(ws, tag + '.fr', S, d); x = self._abuf(ws, tag + '.fx', S, d)
        m = self._buf(ws, tag + '.fm', S); s = self._buf(ws, tag + '.fs', S)
        sh, sf = self._new_site(), self._new_site()
        sites['ffn'] = (sh, sf)
        gam, bet = self.P(pre + 'layer_norm.weight'), self.P(pre + 'layer_norm.bias')
        if self.strip:
            sv = (lambda t: t.data_ptr()) if save else (lambda t: 0)
            self._mlp(plan, ws, 0, S, x_in, key + '.ffn', x.data_ptr(), b1=self.P(pre + 'positionwise_feedforward.fc_1.bias'),
                      b2=self.P(pre + 'positionwise_feedforward.fc_2.bias'), h_out=sv(h), site_h=sh, site_o=sf, ln=(gam, bet, sv(r), sv(m), sv(s)))
            return x.data_ptr()
        self._nt(plan, ws, S, p, d, x_in, d, self.Wp(key + '.f1'), self.P(pre + 'positionwise_feedforward.fc_1.bias'), h.data_ptr(), p,
                 act=1, drop_site=sh, c_bf=True)
        self._nt(plan, ws, S, d, p, h.data_ptr(), p, self.Wp(key + '.f2'), self.P(pre + 'positionwise_feedforward.fc_2.bias'), x.data_ptr(), d,
                 drop_site=sf, residual=x_in, ldr=d, ln=(gam, bet, r.data_ptr(), m.data_ptr(), s.data_ptr()), a_bf=True)
        return x.data_ptr()

    def _build_forward(self, ws, save=True):
        """save=True: the training plan (everything a backward needs is written); save=False (strip mode only): the inference plan --
        no pre-LayerNorm sums, statistics or hidden activations are stored."""
        B, T, F, N, V, d, p = ws['B'], self.T, self.F, self.N, self.V, self.d, self.p
        Se, Sn, BT, BN = B * T * F, B * T * N, B * T, B * N
        plan = []
        ws['sites'] = {}
        st = self.strip                              # strip kernels (bf16 or x3)
        bs = self.bfs                                # ... on a bf16 activation stream
        spec = self._buf(ws, 'spec', B, F, self.W)
        win = self._buf(ws, 'win', Se, self.Kp)
        x0 = self._abuf(ws, 'x0', Se, d)
        e = 'encoder_spec2midi.'
        plan.append(('im2win', (win.data_ptr(), B, F, T, self.n_proc, self.Kp), 'im2win', None))       # (the source pointer is this forward's: ws['spec_ptr'])
        s_emb = self._new_site()
        ws['sites']['embed'] = s_emb
        self._nt(plan, ws, Se, d, self.Kp, win.data_ptr(), self.Kp, self.Wp('embed'), self.Fp('embed_b'), x0.data_ptr(), d,
                 out_scale=math.sqrt(d), add_table=self.P(e + 'pos_embedding_freq.weight'), add_mod=F, drop_site=s_emb, c_bf=bs)
        x = x0.data_ptr()
        ws['enc_in'] = [x]
        for i in range(self.Le):
            x = self._enc_layer_fwd(plan, ws, f'enc{i}', f'enc{i}', f'{e}layers_freq.{i}.', Se, BT, F, self.He, x, save=save)
            ws['enc_in'].append(x)
        enc = x
        # ---------------- decoder, frequency axis (cross attention notes x bins) ----------------
        dd = 'decoder_spec2midi.'
        H = self.Hd
        pos_dec = self.P(dd + 'pos_embedding_freq.weight')
        hz = 2 if self.sb else 4
        q0 = self._buf(ws, 'dec0.q0', N, d, half=True)
        trg = None
        ws['dec_out'] = []
        for j in range(self.Ld):
            tag = f'dec{j}'
            sites = {}
            ws['sites'][tag] = sites
            pre = dd + ('layer_zero_freq.' if j == 0 else f'layers_freq.{j - 1}.')
            gam, bet = self.P(pre + 'layer_norm.weight'), self.P(pre + 'layer_norm.bias')
            sv = (lambda t: t.data_ptr()) if save else (lambda t: 0)
            if j > 0:
                sqkv = self._buf(ws, tag + '.sqkv', Sn, 3 * d, half=True)
                sctx = self._buf(ws, tag + '.sctx', Sn, d, half=True)
                slse = self._buf(ws, tag + '.slse', BT * H * N * 2)
                sr = self._pbuf(ws, tag + '.sr', Sn, d); sx = self._abuf(ws, tag + '.sx', Sn, d)
                sm = self._buf(ws, tag + '.sm', Sn); ss = self._buf(ws, tag + '.ss', Sn)
                s_a, s_o = self._new_site(), self._new_site()
                sites['self'] = (s_a, s_o)
                pls = st and self._planes(H, N, N)
                if st:
                    self._sl(plan, ws, Sn, 3 * d, d, trg, d, tag + '.sa.qkv', self.Fp(tag + '.sa.qkv_b'), sqkv.data_ptr(), 3 * d, c_planes=pls)
                else:
                    self._nt(plan, ws, Sn, 3 * d, d, trg, d, self.Wp(tag + '.sa.qkv'), self.Fp(tag + '.sa.qkv_b'), sqkv.data_ptr(), 3 * d, c_bf=True)
                q = sqkv.data_ptr()
                self._attn(plan, ws, False, BT, H, N, N, q, N * 3 * d, 3 * d, q + hz * d, N * 3 * d, 3 * d, q + 2 * hz * d, N * 3 * d, 3 * d,
                           sctx.data_ptr(), N * d, d, slse.data_ptr(), drop_site=s_a, flags=1 | 2 | 4, planes=pls)
                cq = self._buf(ws, tag + '.cq', Sn, d, half=True)
                if st:
                    self._sl(plan, ws, Sn, d, d, sctx.data_ptr(), d, tag + '.sa.o', self.P(pre + 'self_attention.fc_o.bias'), sx.data_ptr(), d,
                             drop_site=s_o, residual=trg, ldr=d, ln=(gam, bet, sv(sr), sv(sm), sv(ss)))
                    self._sl(plan, ws, Sn, d, d, sx.data_ptr(), d, tag + '.ca.q', self.P(pre + 'encoder_attention.fc_q.bias'), cq.data_ptr(), d,
                             c_planes=self._planes(H, N, F))
                else:
                    self._nt(plan, ws, Sn, d, d, sctx.data_ptr(), d, self.Wp(tag + '.sa.o'), self.P(pre + 'self_attention.fc_o.bias'), sx.data_ptr(), d,
                             drop_site=s_o, residual=trg, ldr=d, ln=(gam, bet, sr.data_ptr(), sm.data_ptr(), ss.data_ptr()), a_bf=True)
                    self._nt(plan, ws, Sn, d, d, sx.data_ptr(), d, self.Wp(tag + '.ca.q'), self.P(pre + 'encoder_attention.fc_q.bias'), cq.data_ptr(), d, c_bf=True)
                cross_in = sx.data_ptr()
                qaddr, qss = cq.data_ptr(), N * d
                res, res_mod = cross_in, 0
            else:
                self._nt(plan, ws, N, d, d, pos_dec, d, self.Wp(tag + '.ca.q'), self.P(pre + 'encoder_attention.fc_q.bias'), q0.data_ptr(), d, c_bf=True)
                qaddr, qss = q0.data_ptr(), 0
                if st and self._planes(H, N, F):     # the shared query of layer zero comes from the block GEMM as fp32: one small conversion
                    q0p = self._buf(ws, 'dec0.q0p', N, d)
                    plan.append((self.lib.hftt_x3_to_planes, (q0.data_ptr(), d, q0p.data_ptr(), d, N, d), 'x3_to_planes', None))
                    qaddr = q0p.data_ptr()
                res, res_mod = (self.wbf.data_ptr() + 2 * self.Woff['dec_pos_bf'] if bs else pos_dec), N
            merged = st and getattr(self, 'merge_ckv', False)
            ldkv = (self.Ld if merged else 1) * 2 * d
            if merged:
                ckv = self._buf(ws, 'dec.ckv_all', Se, ldkv)
                if j == 0:                            # one projection for every layer's K / V (the layers' column blocks of one [Se, Ld * 2d] plane tensor)
                    self._sl(plan, ws, Se, ldkv, d, enc, d, 'dec.ca.kv_all', self.Fp('dec.ca.kv_all_b'), ckv.data_ptr(), ldkv, c_planes=True)
            else:
                ckv = self._buf(ws, tag + '.ckv', Se, 2 * d, half=True)
            cctx = self._buf(ws, tag + '.cctx', Sn, d, half=True)
            clse = self._buf(ws, tag + '.clse', BT * H * N * 2)
            cr = self._pbuf(ws, tag + '.cr', Sn, d); cx = self._abuf(ws, tag + '.cx', Sn, d)
            cm = self._buf(ws, tag + '.cm', Sn); cs = self._buf(ws, tag + '.cs', Sn)
            c_a, c_o = self._new_site(), self._new_site()
            sites['cross'] = (c_a, c_o)
            plc = st and self._planes(H, N, F)
            if merged:
                pass
            elif st:
                self._sl(plan, ws, Se, 2 * d, d, enc, d, tag + '.ca.kv', self.Fp(tag + '.ca.kv_b'), ckv.data_ptr(), 2 * d, c_planes=plc)
            else:
                self._nt(plan, ws, Se, 2 * d, d, enc, d, self.Wp(tag + '.ca.kv'), self.Fp(tag + '.ca.kv_b'), ckv.data_ptr(), 2 * d, c_bf=True)
            kk = ckv.data_ptr() + (j * 2 * d * hz if merged else 0)
            ws.setdefault('ckv_at', {})[tag] = (kk, ldkv)
            ad = self._attn(plan, ws, False, BT, H, N, F, qaddr, qss, d, kk, F * ldkv, ldkv, kk + hz * d, F * ldkv, ldkv,
                            cctx.data_ptr(), N * d, d, clse.data_ptr(), drop_site=c_a, flags=1 | 2 | 4, planes=plc, map_out=(j == self.Ld - 1))
            if j == self.Ld - 1:
                ws.setdefault('attn_out_descs', []).append(ad)
            if st:
                self._sl(plan, ws, Sn, d, d, cctx.data_ptr(), d, tag + '.ca.o', self.P(pre + 'encoder_attention.fc_o.bias'), cx.data_ptr(), d,
                         drop_site=c_o, residual=res, ldr=d, res_mod=res_mod, ln=(gam, bet, sv(cr), sv(cm), sv(cs)))
            else:
                self._nt(plan, ws, Sn, d, d, cctx.data_ptr(), d, self.Wp(tag + '.ca.o'), self.P(pre + 'encoder_attention.fc_o.bias'), cx.data_ptr(), d,
                         drop_site=c_o, residual=res, ldr=d, res_mod=res_mod, ln=(gam, bet, cr.data_ptr(), cm.data_ptr(), cs.data_ptr()), a_bf=True)
            trg = self._ffn_fwd(plan, ws, tag, tag, pre, Sn, cx.data_ptr(), sites, save=save)
            ws['dec_out'].append(trg)
        # ---------------- heads A ----------------
        logits_f = self._buf(ws, 'logits_f', Sn, self.NHp)
        self._nt(plan, ws, Sn, self.NH, d, trg, d, self.Wp('heads_f'), self.Fp('heads_f_b'), logits_f.data_ptr(), self.NHp, a_bf=bs)
        plan.append(('heads', (logits_f.data_ptr(), 0), 'heads_split', None))
        # ---------------- decoder, time axis ----------------
        y0 = self._abuf(ws, 'y0', Sn, d)
        s_t = self._new_site()
        ws['sites']['time_embed'] = s_t
        plan.append(('time_embed', (trg, self.P(dd + 'pos_embedding_time.weight'), y0.data_ptr(), s_t, 3 if bs else 0), 'time_embed_fwd', None))
        y = y0.data_ptr()
        ws['time_in'] = [y]
        for i in range(self.Ld):
            y = self._enc_layer_fwd(plan, ws, f'time{i}', f'time{i}', f'{dd}layers_time.{i}.', Sn, BN, T, H, y, save=save)
            ws['time_in'].append(y)
        logits_t = self._buf(ws, 'logits_t', Sn, self.NHp)
        self._nt(plan, ws, Sn, self.NH, d, y, d, self.Wp('heads_t'), self.Fp('heads_t_b'), logits_t.data_ptr(), self.NHp, a_bf=bs)
        plan.append(('heads', (logits_t.data_ptr(), 1), 'heads_split', None))
        if self.x3 and self.strip and not self.strip_small and (self.fuse_offn_opt == 'all' or (self.fuse_offn_opt == 'inference' and not save)):
            plan = self._fuse_attn_out_ffn(plan, save)
        ws['fwd' if save else 'fwd_inf'] = plan
        ws['enc'] = enc

    def _fuse_attn_out_ffn(self, plan, save):
        """Peephole over a forward plan: hftt_strip_linear (fc_o + dropout + residual + LayerNorm, 256 -> 256) directly followed by the
        hftt_ffn_res_ln_fwd that reads its output becomes ONE hftt_attn_out_ffn_fwd launch on the same two descriptors (their dropout sites, saved
        tensors and statistics unchanged); in the inference plan the LayerNorm-1 output is not written at all."""
        out, i = [], 0
        while i < len(plan):
            e = plan[i]
            nx = plan[i + 1] if i + 1 < len(plan) else None
            if nx is not None and e[2] == 'strip_linear' and nx[2] == 'ffn_fwd':
                o, f = e[1][0]._obj, nx[1][0]._obj
                if (o.ln_gamma and o.N == 256 and o.K == 256 and o.residual and f.mode == 0 and f.d == 256 and f.p == 512 and f.x == o.C
                        and f.w == o.w + 2 * 2 * 256 * 256 and not o.gate and not f.residual):
                    if not save:
                        o.C = 0                           # (x1 lives in registers only)
                    mo, mf = e[3], nx[3]
                    M = o.M
                    meta = {'kernel': 'x3_oln_mlp_kernel<%s>' % ('true' if self.hh else 'false'), 'flops': mo['flops'] + mf['flops'],
                            'bytes': mo['bytes'] + mf['bytes'] - 4.0 * M * 256 * (1 if save else 2), 'shape': (M, 256, 512), 'saves': mf.get('saves', False),
                            'fused': ('strip_linear', 'ffn_fwd'), 'ffn_flops': mf['flops']}
                    out.append((self.lib.hftt_attn_out_ffn_fwd, (e[1][0], nx[1][0]), 'ffn_fwd', meta))
                    i += 2
                    continue
            out.append(e)
            i += 1
        return out

    # ---- backward of one EncoderLayer; dx_out lives in GA on entry (grad of the layer output) and on exit (grad of input)
    def _enc_layer_bwd(self, plan, ws, tag, key, pre, S, n_seq, L, H, x_in, G, extra_dx=0, in_bf=False, gbf=False, out_bf=False):
        """in_bf / out_bf: the gradient stream GA arrives / leaves stored as bf16; gbf: GA and GB are bf16 inside the layer."""
        d, p = self.d, self.p
        b = ws['bufs']
        sa, so, sh, sf = ws['sites'][tag]
        GA, GB, GC, Gh, Gq, Gx = G              # fp32: GA (stream), GB (dr);  "half": GC (dropped dr), Gh (dh), Gq (dqkv), Gx (dctx)
        hz = 2 if self.sb else 4
        gam = self.P(pre + 'layer_norm.weight')
        dgam, dbet = self.G(pre + 'layer_norm.weight'), self.G(pre + 'layer_norm.bias')
        pf = pre + 'positionwise_feedforward.'
        pa = pre + 'self_attention.'
        use_drop = self.dropout > 0.0
        # LN2 backward
        self._lnb(plan, ws, S, GA, b[tag + '.r2'].data_ptr(), b[tag + '.m2'].data_ptr(), b[tag + '.s2'].data_ptr(), gam,
                  GB, GC if use_drop else 0, sf, dgam, dbet, 0.0, dy_bf=in_bf, dr_bf=gbf)
        dbr, dbr_bf = (GC, True) if use_drop else (GB, gbf)
        # fc_2
        self._tn(plan, ws, S, d, p, dbr, d, b[tag + '.h'].data_ptr(), p, [(0, d, self.G(pf + 'fc_2.weight'), self.G(pf + 'fc_2.bias'))],
                 dy_bf=dbr_bf, x_bf=True)
        self._nt(plan, ws, S, p, d, dbr, d, self.Wp(key + '.f2_t'), 0, Gh, p, gate=b[tag + '.h'].data_ptr(), ldg=p,
                 gate_scale=('inv_keep',), a_bf=dbr_bf, c_bf=True, gate_bf=True)
        # fc_1
        self._tn(plan, ws, S, p, d, Gh, p, b[tag + '.x1'].data_ptr(), d, [(0, p, self.G(pf + 'fc_1.weight'), self.G(pf + 'fc_1.bias'))], dy_bf=True, x_bf=self.strip)
        self._nt(plan, ws, S, d, p, Gh, p, self.Wp(key + '.f1_t'), 0, GA, d, residual=GB, ldr=d, a_bf=True, res_bf=gbf, c_bf=gbf)
        # LN1 backward
        self._lnb(plan, ws, S, GA, b[tag + '.r1'].data_ptr(), b[tag + '.m1'].data_ptr(), b[tag + '.s1'].data_ptr(), gam,
                  GB, GC if use_drop else 0, so, dgam, dbet, 1.0, dy_bf=gbf, dr_bf=gbf)
        # fc_o
        self._tn(plan, ws, S, d, d, dbr, d, b[tag + '.ctx'].data_ptr(), d, [(0, d, self.G(pa + 'fc_o.weight'), self.G(pa + 'fc_o.bias'))],
                 dy_bf=dbr_bf, x_bf=True)
        self._nt(plan, ws, S, d, d, dbr, d, self.Wp(key + '.sa.o_t'), 0, Gx, d, a_bf=dbr_bf, c_bf=True)
        # attention
        qkv = b[tag + '.qkv'].data_ptr()
        self._attn(plan, ws, True, n_seq, H, L, L, qkv, L * 3 * d, 3 * d, qkv + hz * d, L * 3 * d, 3 * d, qkv + 2 * hz * d, L * 3 * d, 3 * d,
                   b[tag + '.ctx'].data_ptr(), L * d, d, b[tag + '.lse'].data_ptr(), drop_site=sa, dout=Gx,
                   dq=Gq, dqss=L * 3 * d, lddq=3 * d, dk=Gq + hz * d, dkss=L * 3 * d, lddk=3 * d, dv=Gq + 2 * hz * d, dvss=L * 3 * d, lddv=3 * d,
                   flags=1 | 2 | 4 | 8 | 16)
        # qkv projection
        self._tn(plan, ws, S, 3 * d, d, Gq, 3 * d, x_in, d,
                 [(0, d, self.G(pa + 'fc_q.weight'), self.G(pa + 'fc_q.bias')), (d, d, self.G(pa + 'fc_k.weight'), self.G(pa + 'fc_k.bias')),
                  (2 * d, d, self.G(pa + 'fc_v.weight'), self.G(pa + 'fc_v.bias'))], dy_bf=True, x_bf=self.strip)
        self._nt(plan, ws, S, d, 3 * d, Gq, 3 * d, self.Wp(key + '.sa.qkv_t'), 0, GA, d, residual=GB, ldr=d, a_bf=True, res_bf=gbf, c_bf=out_bf)

    def _enc_layer_bwd_strip(self, plan, ws, tag, key, pre, S, n_seq, L, H, x_in, G):
        """strip-mode backward of one EncoderLayer: the whole gradient stream (GA, GB, GC, Gh, Gq, Gx) is bf16; the two dX GEMMs of
        the FFN are one fused launch (hftt_ffn_bwd_dx), the others hftt_strip_linear on transposed strip packs."""
        d, p = self.d, self.p
        b = ws['bufs']
        sa, so, sh, sf = ws['sites'][tag]
        GA, GB, GC, Gh, Gq, Gx = G
        gam = self.P(pre + 'layer_norm.weight')
        dgam, dbet = self.G(pre + 'layer_norm.weight'), self.G(pre + 'layer_norm.bias')
        pf = pre + 'positionwise_feedforward.'
        pa = pre + 'self_attention.'
        use_drop = self.dropout > 0.0
        mic = self._mic()                           # the consumers of dr apply the dropout mask themselves: no masked copy GC
        dbr = GB if (mic or not use_drop) else GC
        # LN2 -> FFN
        self._lnb(plan, ws, S, GA, b[tag + '.r2'].data_ptr(), b[tag + '.m2'].data_ptr(), b[tag + '.s2'].data_ptr(), gam,
                  GB, GC if (use_drop and not mic) else 0, sf, dgam, dbet, 0.0, dy_bf=True, dr_bf=True)
        self._tn(plan, ws, S, d, p, dbr, d, b[tag + '.h'].data_ptr(), p, [(0, d, self.G(pf + 'fc_2.weight'), self.G(pf + 'fc_2.bias'))], dy_bf=True, x_bf=True, x_hid=True,
                 dy_drop_site=sf if mic else 0)
        self._mlp(plan, ws, 1, S, dbr, key + '.ffn_t', GA, h_out=Gh, gate=b[tag + '.h'].data_ptr(), gate_scale=('inv_keep',), residual=GB, site_o=sf if mic else 0)
        self._tn(plan, ws, S, p, d, Gh, p, b[tag + '.x1'].data_ptr(), d, [(0, p, self.G(pf + 'fc_1.weight'), self.G(pf + 'fc_1.bias'))], dy_bf=True, x_bf=True, dy_hid=True)
        # LN1 -> attention
        self._lnb(plan, ws, S, GA, b[tag + '.r1'].data_ptr(), b[tag + '.m1'].data_ptr(), b[tag + '.s1'].data_ptr(), gam,
                  GB, GC if (use_drop and not mic) else 0, so, dgam, dbet, 1.0, dy_bf=True, dr_bf=True)
        self._tn(plan, ws, S, d, d, dbr, d, b[tag + '.ctx'].data_ptr(), d, [(0, d, self.G(pa + 'fc_o.weight'), self.G(pa + 'fc_o.bias'))], dy_bf=True, x_bf=True,
                 dy_drop_site=so if mic else 0)
        self._sl(plan, ws, S, d, d, dbr, d, key + '.sa.o_t', 0, Gx, d, x_drop_site=so if mic else 0)
        qkv = b[tag + '.qkv'].data_ptr()
        hz = 2 if self.sb else 4
        self._attn(plan, ws, True, n_seq, H, L, L, qkv, L * 3 * d, 3 * d, qkv + hz * d, L * 3 * d, 3 * d, qkv + 2 * hz * d, L * 3 * d, 3 * d,
                   b[tag + '.ctx'].data_ptr(), L * d, d, b[tag + '.lse'].data_ptr(), drop_site=sa, dout=Gx,
                   dq=Gq, dqss=L * 3 * d, lddq=3 * d, dk=Gq + hz * d, dkss=L * 3 * d, lddk=3 * d, dv=Gq + 2 * hz * d, dvss=L * 3 * d, lddv=3 * d,
                   flags=1 | 2 | 4 | 8 | 16, planes=self._planes(H, L, L))
        self._tn(plan, ws, S, 3 * d, d, Gq, 3 * d, x_in, d,
                 [(0, d, self.G(pa + 'fc_q.weight'), self.G(pa + 'fc_q.bias')), (d, d, self.G(pa + 'fc_k.weight'), self.G(pa + 'fc_k.bias')),
                  (2 * d, d, self.G(pa + 'fc_v.weight'), self.G(pa + 'fc_v.bias'))], dy_bf=True, x_bf=True)
        self._sl(plan, ws, S, d, 3 * d, Gq, 3 * d, key + '.sa.qkv_t', 0, GA, d, residual=GB, ldr=d)

    def _ffn_bwd(self, plan, ws, tag, key, pre, S, x_in, G, ln_beta):
        """FFN + LN block of the decoder layers: grad of output in GA -> grad of x_in in GA."""
        d, p = self.d, self.p
        b = ws['bufs']
        sh, sf = ws['sites'][tag]['ffn']
        GA, GB, GC, Gh, Gq, Gx = G
        gam = self.P(pre + 'layer_norm.weight')
        dgam, dbet = self.G(pre + 'layer_norm.weight'), self.G(pre + 'layer_norm.bias')
        pf = pre + 'positionwise_feedforward.'
        use_drop = self.dropout > 0.0
        if self.strip:
            mic = self._mic()
            dbr = GB if (mic or not use_drop) else GC
            self._lnb(plan, ws, S, GA, b[tag + '.fr'].data_ptr(), b[tag + '.fm'].data_ptr(), b[tag + '.fs'].data_ptr(), gam,
                      GB, GC if (use_drop and not mic) else 0, sf, dgam, dbet, ln_beta, dy_bf=True, dr_bf=True)
            self._tn(plan, ws, S, d, p, dbr, d, b[tag + '.h'].data_ptr(), p, [(0, d, self.G(pf + 'fc_2.weight'), self.G(pf + 'fc_2.bias'))], dy_bf=True, x_bf=True, x_hid=True,
                     dy_drop_site=sf if mic else 0)
            self._mlp(plan, ws, 1, S, dbr, key + '.ffn_t', GA, h_out=Gh, gate=b[tag + '.h'].data_ptr(), gate_scale=('inv_keep',), residual=GB, site_o=sf if mic else 0)
            self._tn(plan, ws, S, p, d, Gh, p, x_in, d, [(0, p, self.G(pf + 'fc_1.weight'), self.G(pf + 'fc_1.bias'))], dy_bf=True, x_bf=True, dy_hid=True)
            return
        self._lnb(plan, ws, S, GA, b[tag + '.fr'].data_ptr(), b[tag + '.fm'].data_ptr(), b[tag + '.fs'].data_ptr(), gam,
                  GB, GC if use_drop else 0, sf, dgam, dbet, ln_beta)
        dbr, dbr_bf = (GC, True) if use_drop else (GB, False)
        self._tn(plan, ws, S, d, p, dbr, d, b[tag + '.h'].data_ptr(), p, [(0, d, self.G(pf + 'fc_2.weight'), self.G(pf + 'fc_2.bias'))],
                 dy_bf=dbr_bf, x_bf=True)
        self._nt(plan, ws, S, p, d, dbr, d, self.Wp(key + '.f2_t'), 0, Gh, p, gate=b[tag + '.h'].data_ptr(), ldg=p, gate_scale=('inv_keep',),
                 a_bf=dbr_bf, c_bf=True, gate_bf=True)
        self._tn(plan, ws, S, p, d, Gh, p, x_in, d, [(0, p, self.G(pf + 'fc_1.weight'), self.G(pf + 'fc_1.bias'))], dy_bf=True, x_bf=self.strip)
        self._nt(plan, ws, S, d, p, Gh, p, self.Wp(key + '.f1_t'), 0, GA, d, residual=GB, ldr=d, a_bf=True)

    def _build_backward(self, ws):
        B, T, F, N, V, d, p = ws['B'], self.T, self.F, self.N, self.V, self.d, self.p
        Se, Sn, BT, BN = B * T * F, B * T * N, B * T, B * N
        b = ws['bufs']
        plan = []
        H = self.Hd
        dd = 'decoder_spec2midi.'
        e = 'encoder_spec2midi.'
        use_drop = self.dropout > 0.0
        # bf16 gradient stream on the bin-token (encoder-sized) set: the residual-stream gradient and the LN-backward output stored
        # as bf16 between the decoder's cross-attention and the first encoder layer.  Built, parity-tested and MEASURED (r01, B=8):
        # ln_bwd 90.9 -> 79.5 us, but the dX GEMMs' row-pass epilogue then moves 8 bytes per lane instead of 16 and gets slower
        # (K=512: 253 -> 284 us, K=768: 324 -> 366 us) -- net zero, so it is OFF by default (HFTT_BF16_GRAD=1 enables) until the
        # row pass handles 8 columns per lane for bf16 residual / C.
        # (needs the A-stationary GEMM on every encoder dX: d % 256 == 0, K = ff and 3d <= 768, >= 256 bin tokens)
        st = self.strip                              # strip kernels; bs: on the bf16 stream (then the whole gradient stream is bf16)
        bs = self.bfs
        egb = (not st and self.sb and os.environ.get('HFTT_BF16_GRAD', '0') == '1' and d % 256 == 0 and max(p, 3 * d) <= 768 and Se >= 256)
        ws['bf16_grad'] = bool(egb or bs)
        # gradient scratch: note-token sized and bin-token sized sets
        nGA = self._abuf(ws, 'g.nA', Sn, d).data_ptr(); nGB = self._abuf(ws, 'g.nB', Sn, d).data_ptr()
        hz = 2 if self.sb else 4
        nGC = self._buf(ws, 'g.nC', Sn, d, half=True).data_ptr(); nGD = self._abuf(ws, 'g.nD', Sn, d).data_ptr()
        nGh = self._buf(ws, 'g.nh', Sn, p, half=True, hidden=True).data_ptr(); nGq = self._buf(ws, 'g.nq', Sn, 3 * d, half=True).data_ptr()
        nGx = self._buf(ws, 'g.nx', Sn, d, half=True).data_ptr()
        q1f = self._buf(ws, 'g.q1f', Sn, d).data_ptr() if bs else 0          # layer zero's per-sequence dq stays fp32 (summed over sequences)
        eGA = self._abuf(ws, 'g.eA', Se, d).data_ptr(); eGB = self._abuf(ws, 'g.eB', Se, d).data_ptr()
        eGC = self._buf(ws, 'g.eC', Se, d, half=True).data_ptr()
        eGh = self._buf(ws, 'g.eh', Se, p, half=True, hidden=True).data_ptr(); eGq = self._buf(ws, 'g.eq', Se, 3 * d, half=True).data_ptr()
        eGx = self._buf(ws, 'g.ex', Se, d, half=True).data_ptr()
        dlog = self._buf(ws, 'g.dlog', Sn, self.NHp).data_ptr()
        cs_n = max(F * d, N * d, T * d)
        cs_ws = self._buf(ws, 'g.cs', self.lib.hftt_colsum_ws_bytes(1, cs_n) // 4 + 16).data_ptr()
        dq0s = self._buf(ws, 'g.dq0', N * d).data_ptr()
        # incoming gradients of the 8 differentiable outputs (filled by the loss kernel or by autograd glue)
        for nm in ('onset_A', 'offset_A', 'mpe_A', 'onset_B', 'offset_B', 'mpe_B'):
            self._buf(ws, 'd.' + nm, Sn)
        self._buf(ws, 'd.velocity_A', Sn, V); self._buf(ws, 'd.velocity_B', Sn, V)

        def head_segs(tag):
            return [(0, V, self.G(f'{dd}fc_velocity_{tag}.weight'), self.G(f'{dd}fc_velocity_{tag}.bias')),
                    (V, 1, self.G(f'{dd}fc_onset_{tag}.weight'), self.G(f'{dd}fc_onset_{tag}.bias')),
                    (V + 1, 1, self.G(f'{dd}fc_offset_{tag}.weight'), self.G(f'{dd}fc_offset_{tag}.bias')),
                    (V + 2, 1, self.G(f'{dd}fc_mpe_{tag}.weight'), self.G(f'{dd}fc_mpe_{tag}.bias'))]

        # ---- heads B + time layers ----
        plan.append(('heads_bwd', ('B', dlog, 1), 'heads_split_bwd', None))
        y_last = ws['time_in'][-1]
        self._tn(plan, ws, Sn, self.NHp, d, dlog, self.NHp, y_last, d, head_segs('time'), x_bf=self.strip)
        self._nt(plan, ws, Sn, d, self.NHp, dlog, self.NHp, self.Wp('heads_t_t'), 0, nGA, d, c_bf=bs)
        Gn = (nGA, nGB, nGC, nGh, nGq, nGx)
        for i in reversed(range(self.Ld)):
            if st:
                self._enc_layer_bwd_strip(plan, ws, f'time{i}', f'time{i}', f'{dd}layers_time.{i}.', Sn, BN, T, H, ws['time_in'][i], Gn)
            else:
                self._enc_layer_bwd(plan, ws, f'time{i}', f'time{i}', f'{dd}layers_time.{i}.', Sn, BN, T, H, ws['time_in'][i], Gn)
        # ---- heads A, then the time-embedding transpose back onto the note-major gradient ----
        plan.append(('heads_bwd', ('A', dlog, 0), 'heads_split_bwd', None))
        f_last = ws['dec_out'][-1]
        self._tn(plan, ws, Sn, self.NHp, d, dlog, self.NHp, f_last, d, head_segs('freq'), x_bf=self.strip)
        self._nt(plan, ws, Sn, d, self.NHp, dlog, self.NHp, self.Wp('heads_f_t'), 0, nGD, d, c_bf=bs)
        plan.append(('time_embed_bwd', (nGA, nGD, nGB if use_drop else 0, ws['sites']['time_embed'], 7 if bs else 0), 'time_embed_bwd', None))
        plan.append(('colsum', (nGB if use_drop else nGA, BN, T * d, T * d, self.G(dd + 'pos_embedding_time.weight'), 0.0, cs_ws, 1 if bs else 0), 'colsum', None))
        # gradient buckets in the order they become final (flat ranges are contiguous: state_dict order is encoder,
        # frequency decoder + heads A, time decoder + heads B): (plan length when final, flat lo, flat hi)
        o_dec, o_time, o_end = self.poff[dd + 'pos_embedding_freq.weight'], self.poff[dd + 'pos_embedding_time.weight'], self.flat_grads.numel()
        marks = [(len(plan), o_time, o_end)]
        # ---- frequency decoder layers, last to first.  Gradient stream lives in A (= nGD), per-sequence dq in Q1 (= nGA);
        #      the encoder-output gradient accumulates in eGA ----
        A, Bf, Cf, Q1 = nGD, nGB, nGC, nGA
        Gd = (A, Bf, Cf, nGh, nGq, nGx)
        mic = st and self._mic()                      # (the LayerNorm backward writes no masked copy: its consumers mask dr themselves)
        dbr, dbr_bf = (Cf, True) if (use_drop and not mic) else (Bf, bs)
        first_enc_grad = True
        enc = ws['enc']
        for j in reversed(range(self.Ld)):
            tag = f'dec{j}'
            pre = dd + ('layer_zero_freq.' if j == 0 else f'layers_freq.{j - 1}.')
            sites = ws['sites'][tag]
            gam = self.P(pre + 'layer_norm.weight')
            dgam, dbet = self.G(pre + 'layer_norm.weight'), self.G(pre + 'layer_norm.bias')
            pc = pre + 'encoder_attention.'
            self._ffn_bwd(plan, ws, tag, tag, pre, Sn, b[tag + '.cx'].data_ptr(), Gd, 0.0)
            c_a, c_o = sites['cross']
            self._lnb(plan, ws, Sn, A, b[tag + '.cr'].data_ptr(), b[tag + '.cm'].data_ptr(), b[tag + '.cs'].data_ptr(), gam,
                      Bf, Cf if (use_drop and not mic) else 0, c_o, dgam, dbet, 1.0, dy_bf=st, dr_bf=st)
            self._tn(plan, ws, Sn, d, d, dbr, d, b[tag + '.cctx'].data_ptr(), d, [(0, d, self.G(pc + 'fc_o.weight'), self.G(pc + 'fc_o.bias'))],
                     dy_bf=dbr_bf, x_bf=True, dy_drop_site=c_o if mic else 0)
            if st:
                self._sl(plan, ws, Sn, d, d, dbr, d, tag + '.ca.o_t', 0, nGx, d, x_drop_site=c_o if mic else 0)
            else:
                self._nt(plan, ws, Sn, d, d, dbr, d, self.Wp(tag + '.ca.o_t'), 0, nGx, d, a_bf=dbr_bf, c_bf=True)
            kk, ldkv = ws['ckv_at'][tag]
            plc = st and self._planes(H, N, F)
            if j > 0:
                qaddr, qss = b[tag + '.cq'].data_ptr(), N * d
            else:
                qaddr, qss = b['dec0.q0p' if plc else 'dec0.q0'].data_ptr(), 0
            # dq (per sequence) -> Q1 ; dk,dv -> eGq viewed as [Se, 2d]
            # per-sequence dq stays fp32 (layer zero sums it over sequences with the fp32 colsum); dk, dv are "half" tensors
            # strip mode: dq of the layers with their own query projection is a GEMM operand only -> bf16; layer zero keeps fp32 (q1f)
            dq_buf = Q1 if not bs else (Q1 if j > 0 else q1f)
            # merged (x3 strip plans, three decoder layers): dk / dv of every layer go into the column blocks of ONE [Se, 6d] tensor; the weight
            # gradients (one product, six segments) and the encoder-output gradient (two K = 768 halves) are formed once, behind layer zero --
            # the encoder output is read once instead of three times, the accumulating gradient makes one round trip less
            mb = st and getattr(self, 'merge_ckv_bwd', False)
            if mb:
                gkv = self._buf(ws, 'g.ekv_all', Se, self.Ld * 2 * d).data_ptr() + j * 2 * d * 4
                ldg = self.Ld * 2 * d
            else:
                gkv, ldg = eGq, 2 * d
            self._attn(plan, ws, True, BT, H, N, F, qaddr, qss, d, kk, F * ldkv, ldkv, kk + hz * d, F * ldkv, ldkv,
                       b[tag + '.cctx'].data_ptr(), N * d, d, b[tag + '.clse'].data_ptr(), drop_site=c_a, dout=nGx,
                       dq=dq_buf, dqss=N * d, lddq=d, dk=gkv, dkss=F * ldg, lddk=ldg, dv=gkv + hz * d, dvss=F * ldg, lddv=ldg,
                       flags=1 | 2 | 4 | 16 | (8 if (st and j > 0) else 0), planes=plc)
            if mb:
                if j == 0:
                    gall = ws['bufs']['g.ekv_all'].data_ptr()
                    segs = []
                    for jj in range(self.Ld):
                        pcj = dd + ('layer_zero_freq.' if jj == 0 else f'layers_freq.{jj - 1}.') + 'encoder_attention.'
                        segs += [(2 * jj * d, d, self.G(pcj + 'fc_k.weight'), self.G(pcj + 'fc_k.bias')), ((2 * jj + 1) * d, d, self.G(pcj + 'fc_v.weight'), self.G(pcj + 'fc_v.bias'))]
                    self._tn(plan, ws, Se, ldg, d, gall, ldg, enc, d, segs, dy_bf=True, x_bf=self.strip)
                    self._sl(plan, ws, Se, d, 3 * d, gall, ldg, 'dec.ca.kv_all_t0', 0, eGA, d)
                    self._sl(plan, ws, Se, d, 3 * d, gall + 3 * d * 4, ldg, 'dec.ca.kv_all_t1', 0, eGA, d, residual=eGA, ldr=d)
                    first_enc_grad = False
            else:
                self._tn(plan, ws, Se, 2 * d, d, eGq, 2 * d, enc, d,
                         [(0, d, self.G(pc + 'fc_k.weight'), self.G(pc + 'fc_k.bias')), (d, d, self.G(pc + 'fc_v.weight'), self.G(pc + 'fc_v.bias'))], dy_bf=True, x_bf=self.strip)
            if mb:
                pass
            elif st:                                 # (in place: a lane reads exactly the residual elements it then overwrites)
                self._sl(plan, ws, Se, d, 2 * d, eGq, 2 * d, tag + '.ca.kv_t', 0, eGA, d, residual=0 if first_enc_grad else eGA, ldr=d)
                first_enc_grad = False
            elif first_enc_grad:
                self._nt(plan, ws, Se, d, 2 * d, eGq, 2 * d, self.Wp(tag + '.ca.kv_t'), 0, eGA, d, a_bf=True, c_bf=egb)
                first_enc_grad = False
            else:
                self._nt(plan, ws, Se, d, 2 * d, eGq, 2 * d, self.Wp(tag + '.ca.kv_t'), 0, eGA, d, residual=eGA, ldr=d, a_bf=True, res_bf=egb, c_bf=egb)
            if j > 0:
                # q projection of the cross attention (input sx, which is also the residual of this block)
                self._tn(plan, ws, Sn, d, d, Q1, d, b[tag + '.sx'].data_ptr(), d, [(0, d, self.G(pc + 'fc_q.weight'), self.G(pc + 'fc_q.bias'))], dy_bf=st, x_bf=st)
                if st:
                    self._sl(plan, ws, Sn, d, d, Q1, d, tag + '.ca.q_t', 0, A, d, residual=Bf, ldr=d)
                else:
                    self._nt(plan, ws, Sn, d, d, Q1, d, self.Wp(tag + '.ca.q_t'), 0, A, d, residual=Bf, ldr=d)
                # self-attention block (input trg = previous layer output)
                s_a, s_o = sites['self']
                ps = pre + 'self_attention.'
                trg = ws['dec_out'][j - 1]
                self._lnb(plan, ws, Sn, A, b[tag + '.sr'].data_ptr(), b[tag + '.sm'].data_ptr(), b[tag + '.ss'].data_ptr(), gam,
                          Bf, Cf if (use_drop and not mic) else 0, s_o, dgam, dbet, 1.0, dy_bf=st, dr_bf=st)
                self._tn(plan, ws, Sn, d, d, dbr, d, b[tag + '.sctx'].data_ptr(), d, [(0, d, self.G(ps + 'fc_o.weight'), self.G(ps + 'fc_o.bias'))],
                         dy_bf=dbr_bf, x_bf=True, dy_drop_site=s_o if mic else 0)
                if st:
                    self._sl(plan, ws, Sn, d, d, dbr, d, tag + '.sa.o_t', 0, nGx, d, x_drop_site=s_o if mic else 0)
                else:
                    self._nt(plan, ws, Sn, d, d, dbr, d, self.Wp(tag + '.sa.o_t'), 0, nGx, d, a_bf=dbr_bf, c_bf=True)
                q = b[tag + '.sqkv'].data_ptr()
                self._attn(plan, ws, True, BT, H, N, N, q, N * 3 * d, 3 * d, q + hz * d, N * 3 * d, 3 * d, q + 2 * hz * d, N * 3 * d, 3 * d,
                           b[tag + '.sctx'].data_ptr(), N * d, d, b[tag + '.slse'].data_ptr(), drop_site=s_a, dout=nGx,
                           dq=nGq, dqss=N * 3 * d, lddq=3 * d, dk=nGq + hz * d, dkss=N * 3 * d, lddk=3 * d, dv=nGq + 2 * hz * d, dvss=N * 3 * d, lddv=3 * d,
                           flags=1 | 2 | 4 | 8 | 16, planes=st and self._planes(H, N, N))
                self._tn(plan, ws, Sn, 3 * d, d, nGq, 3 * d, trg, d,
                         [(0, d, self.G(ps + 'fc_q.weight'), self.G(ps + 'fc_q.bias')), (d, d, self.G(ps + 'fc_k.weight'), self.G(ps + 'fc_k.bias')),
                          (2 * d, d, self.G(ps + 'fc_v.weight'), self.G(ps + 'fc_v.bias'))], dy_bf=True, x_bf=self.strip)
                if st:
                    self._sl(plan, ws, Sn, d, 3 * d, nGq, 3 * d, tag + '.sa.qkv_t', 0, A, d, residual=Bf, ldr=d)
                else:
                    self._nt(plan, ws, Sn, d, 3 * d, nGq, 3 * d, self.Wp(tag + '.sa.qkv_t'), 0, A, d, residual=Bf, ldr=d, a_bf=True)
            else:
                # layer zero: query = fc_q(pos_embedding_freq) shared by all sequences, residual = pos_embedding_freq
                gpos = self.G(dd + 'pos_embedding_freq.weight')
                plan.append(('colsum', (Bf, BT, N * d, N * d, gpos, 0.0, cs_ws, 1 if bs else 0), 'colsum', None))       # residual path (undropped dr)
                plan.append(('colsum', (q1f if bs else Q1, BT, N * d, N * d, dq0s, 0.0, cs_ws, 0), 'colsum', None))     # sum of per-sequence dq
                self._tn(plan, ws, N, d, d, dq0s, d, self.P(dd + 'pos_embedding_freq.weight'), d,
                         [(0, d, self.G(pc + 'fc_q.weight'), self.G(pc + 'fc_q.bias'))])
                self._nt(plan, ws, N, d, d, dq0s, d, self.Wp(tag + '.ca.q_t'), 0, gpos, d, residual=gpos, ldr=d)
        marks.append((len(plan), o_dec, o_time))
        # ---- encoder layers ----
        Ge = (eGA, eGB, eGC, eGh, eGq, eGx)
        for i in reversed(range(self.Le)):
            if st:
                self._enc_layer_bwd_strip(plan, ws, f'enc{i}', f'enc{i}', f'{e}layers_freq.{i}.', Se, BT, F, self.He, ws['enc_in'][i], Ge)
                continue
            self._enc_layer_bwd(plan, ws, f'enc{i}', f'enc{i}', f'{e}layers_freq.{i}.', Se, BT, F, self.He, ws['enc_in'][i], Ge,
                                in_bf=egb, gbf=egb, out_bf=(egb and i > 0))     # the embedding stage below reads fp32
        # ---- embedding ----
        plan.append(('dropout_bwd', (eGA, Se * d, ws['sites']['embed'], 1 if bs else 0), 'dropout_bwd', None))
        plan.append(('colsum', (eGA, BT, F * d, F * d, self.G(e + 'pos_embedding_freq.weight'), 0.0, cs_ws, 1 if bs else 0), 'colsum', None))
        self._tn(plan, ws, Se, d, self.Kp, eGA, d, b['win'].data_ptr(), self.Kp, [(0, d, self.dweff.data_ptr(), self.dbeff.data_ptr())],
                 out_scale=math.sqrt(d), dy_bf=st)
        plan.append((self.lib.hftt_embed_fold_bwd, (C.byref(self.fold),), 'embed_fold_bwd', None))
        marks.append((len(plan), 0, o_dec))
        ws['bwd'] = plan
        ws['bwd_marks'] = marks

    # ------------------------------------------------------------------ running plans
    def _run(self, ws, plan, stream, outs=None, seed=0, p=0.0):
        L = self.lib
        B, T, N, V, d = ws['B'], self.T, self.N, self.V, self.d
        prof = self.profiler
        for fn, args, name, meta in plan:
            if prof is not None:
                prof.begin(name, meta)
            if not isinstance(fn, str):
                rc = fn(*args, stream)
            elif fn == 'im2win':
                rc = L.hftt_im2win(ws['spec_ptr'], *args, stream)
            elif fn == 'ln_reduce':
                n_wg, n, dg, db, beta = args
                rc = L.hftt_ln_bwd_reduce(ws['ln_ws_ptr'], n_wg, n, dg, db, beta, stream)
            elif fn == 'heads':
                logits, tm = args
                o = outs[5:9] if tm else outs[0:4]
                rc = L.hftt_heads_split(logits, self.NHp, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), B, T, N, V, tm, stream)
            elif fn == 'heads_bwd':
                side, dlog, tm = args
                b = ws['bufs']
                po = outs[5:8] if tm else outs[0:3]
                rc = L.hftt_heads_split_bwd(po[0].data_ptr(), po[1].data_ptr(), po[2].data_ptr(),
                                            b['d.onset_' + side].data_ptr(), b['d.offset_' + side].data_ptr(), b['d.mpe_' + side].data_ptr(),
                                            b['d.velocity_' + side].data_ptr(), dlog, self.NHp, B, T, N, V, tm, stream)
            elif fn == 'time_embed':
                x, pos, y, site, iof = args
                rc = L.hftt_time_embed_fwd(x, pos, y, B, T, N, d, math.sqrt(d), p, site, seed, iof, stream)
            elif fn == 'time_embed_bwd':
                dy, dx, dym, site, iof = args
                rc = L.hftt_time_embed_bwd(dy, dx, dym, B, T, N, d, math.sqrt(d), p, site, seed, 1, iof, stream)
            elif fn == 'colsum':
                x, rows, n, ld, out, beta, wsp, xbf = args
                rc = L.hftt_colsum(x, rows, n, ld, out, beta, wsp, xbf, stream)
            elif fn == 'dropout_bwd':
                g, n, site, gbf = args
                rc = L.hftt_dropout_bwd(g, n, p, site, seed, gbf, stream) if p > 0.0 else 0
            else:
                raise _capi.HfttError('unknown plan op %s' % fn)
            if prof is not None:
                prof.end()
            if rc != 0:
                check(rc, name)

    def _patch(self, ws, p, seed):
        inv_keep = keep_scale(p)
        for dsc in ws['drop']:
            dsc.drop_p = p
            dsc.drop_seed = seed
        for dsc in ws.get('gate_descs', ()):
            dsc.gate_scale = inv_keep

    def forward(self, spec, training=False, outputs=None, save=None):
        """spec [B, n_bin, M+T+M] (any device/dtype/strides) -> 9-tuple of fresh fp32 tensors on the engine's device."""
        if spec.dim() != 3 or spec.shape[1] != self.F or spec.shape[2] != self.W:
            raise _capi.HfttError('input_spec must be [B, %d, %d], got %s' % (self.F, self.W, tuple(spec.shape)))
        if not self.is_bound():
            raise _capi.HfttError('engine parameters are not bound (call bind first)')
        B = spec.shape[0]
        with torch.cuda.device(self.device):         # launches, stream lookup and device queries all refer to the engine's GPU
            return self._forward(spec, B, training, outputs, save)

    def _forward(self, spec, B, training, outputs, save):
        ws = self.workspace(B)
        self._cur_ws = ws
        # the window gather reads the caller's tensor in place when it already is what the kernel wants (fp32, contiguous, on this device);
        # anything else is converted into the workspace copy.  (torch's device-to-device copy_ is ~24 blit launches of 64 KB here: 80 us.)
        if spec.is_cuda and spec.device == self.device and spec.dtype == torch.float32 and spec.is_contiguous():
            ws['spec_ptr'], ws['spec_ref'] = spec.data_ptr(), spec
        else:
            ws['bufs']['spec'].copy_(spec)
            ws['spec_ptr'], ws['spec_ref'] = ws['bufs']['spec'].data_ptr(), None
        T, N, V = self.T, self.N, self.V
        dev = self.device
        outs = outputs
        if outs is None:
            outs = [torch.empty(B, T, N, device=dev) for _ in range(3)] + [torch.empty(B, T, N, V, device=dev)] \
                + [torch.empty(B, T, self.Hd, N, self.F, device=dev)] \
                + [torch.empty(B, T, N, device=dev) for _ in range(3)] + [torch.empty(B, T, N, V, device=dev)]
        for ad in ws['attn_out_descs']:
            ad.probs = outs[4].data_ptr()
        p = self.dropout if training else 0.0
        if training:
            self.step_counter += 1
        seed = (self.base_seed * 1000003 + self.step_counter) & 0xFFFFFFFFFFFFFFFF
        ws['seed'], ws['p'], ws['outs'] = seed, p, outs
        self._patch(ws, p, seed)
        stream = torch.cuda.current_stream(dev).cuda_stream
        self.prepare_weights(stream)
        if save is None:
            save = training
        plan = ws['fwd'] if (save or 'fwd_inf' not in ws) else ws['fwd_inf']      # inference plan: nothing is saved for a backward
        ws['saved'] = plan is ws['fwd']
        self._run(ws, plan, stream, outs=outs, seed=seed, p=p)
        self.generation += 1
        ws['generation'] = self.generation
        return tuple(outs)

    def backward(self, B, generation=None, on_ready=None):
        """Backward through the most recent forward of batch size B; the gradients of the 8 outputs must be in
        the 'd.*' workspace buffers.  Fills flat_grads (every parameter written exactly once).
        on_ready(lo, hi), if given, is called three times, each as soon as every launch that writes flat_grads[lo:hi] has
        been enqueued (time decoder, frequency decoder, encoder): the data-parallel all-reduce of that range can then
        overlap the rest of the backward (hftt_hip/ddp.py)."""
        ws = self._ws.get(B)
        if ws is None or 'outs' not in ws:
            raise _capi.HfttError('backward called without a forward')
        if not ws.get('saved', True):
            raise _capi.HfttError('backward after an inference-plan forward (model.eval()): nothing was saved; run the forward in training mode')
        if generation is not None and generation != ws['generation']:
            raise _capi.HfttError('backward called for a stale forward (activations were overwritten by a later forward)')
        self._cur_ws = ws
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self._patch(ws, ws['p'], ws['seed'])
            if on_ready is None:
                self._run(ws, ws['bwd'], stream, outs=ws['outs'], seed=ws['seed'], p=ws['p'])
                return
            start = 0
            for end, lo, hi in ws['bwd_marks']:
                self._run(ws, ws['bwd'][start:end], stream, outs=ws['outs'], seed=ws['seed'], p=ws['p'])
                on_ready(lo, hi)
                start = end

    # ------------------------------------------------------------------ fused loss (training/train.py:141-153)
    def loss(self, B, labels, weight_A=1.0, weight_B=1.0, with_grad=True):
        """labels: (onset, offset, mpe, velocity) device tensors [B,T,N] (fp32, fp32, fp32, int64).
        Returns a [9] device tensor (total + 8 terms); with_grad fills the 'd.*' buffers."""
        ws = self._ws[B]
        outs = ws['outs']
        b = ws['bufs']
        n = B * self.T * self.N
        if 'loss_ws' not in b:
            b['loss_ws'] = torch.empty(self.lib.hftt_loss_ws_bytes(n) // 4 + 16, dtype=torch.float32, device=self.device)
            b['loss_out'] = torch.zeros(16, dtype=torch.float32, device=self.device)
        lo, lf, lm, lv = labels
        for t, dt in ((lo, torch.float32), (lf, torch.float32), (lm, torch.float32), (lv, torch.int64)):
            if t.dtype != dt or not t.is_contiguous() or t.device != self.device or t.numel() != n:
                raise _capi.HfttError('labels must be contiguous device tensors [B,T,N] (fp32,fp32,fp32,int64)')
        dsc = LossDesc()
        dsc.n, dsc.V = n, self.V
        for i, k in enumerate((0, 1, 2, 5, 6, 7)):
            dsc.prob[i] = outs[k].data_ptr()
        dsc.vel[0], dsc.vel[1] = outs[3].data_ptr(), outs[8].data_ptr()
        dsc.label_onset, dsc.label_offset, dsc.label_mpe, dsc.label_velocity = lo.data_ptr(), lf.data_ptr(), lm.data_ptr(), lv.data_ptr()
        dsc.weight_A, dsc.weight_B, dsc.grad_scale = weight_A, weight_B, 1.0
        if with_grad:
            for i, nm in enumerate(('onset_A', 'offset_A', 'mpe_A', 'onset_B', 'offset_B', 'mpe_B')):
                dsc.d_prob[i] = b['d.' + nm].data_ptr()
            dsc.d_vel[0], dsc.d_vel[1] = b['d.velocity_A'].data_ptr(), b['d.velocity_B'].data_ptr()
        dsc.loss_out, dsc.ws = b['loss_out'].data_ptr(), b['loss_ws'].data_ptr()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            check(self.lib.hftt_loss(C.byref(dsc), stream), 'loss')
        return b['loss_out'][:9]
