"""The split-operand ("x3") precision mode: kernels through the C ABI against fp64 references (the whole model in this mode against the
reference-generated golden fixtures at north_star's 1e-3: tests/test_model_gpu.py, precision 'x3').

npass 2 = fp16 hi + lo (forward products), npass 4 = bf16 hi + lo (products with a gradient operand); see csrc/x3_common.h."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util
from util import keep_scale, rel_err, max_err, keep_mask_t

pytestmark = pytest.mark.gpu

# relative to the output's max magnitude.  fp16 pair: 2^-22 operand error, the fp32 accumulation of the MFMA dominates; bf16 pair: 2^-16
TOL = {2: 4e-6, 4: 6e-5}


def _ops():
    from hftt_hip import ops
    return ops


def _grad_hi_built():
    """the opt-in gradient-rounding forms are compiled only with HFTT_BUILD_GRAD_HI=1 (the default library rejects their flags)"""
    from hftt_hip import _capi
    return bool(_capi.lib().hftt_build_options() & 1)


@pytest.mark.parametrize('npass', [2, 4])
def test_split_planes_reconstruct_the_weights(dev, npass):
    """hi + lo of hftt_prep_weights_x3, including values whose lo half is an fp16 subnormal and values beyond fp16's range (clamped)."""
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    w = torch.randn(96, 64, generator=g) * torch.logspace(-6, 3, 64).unsqueeze(0)
    pl = ops.prepare_weight(w.to(dev), npass=npass)
    hi, lo = pl[0, :96].cpu(), pl[1, :96].cpu()
    dt = torch.float16 if npass == 2 else torch.bfloat16
    rec = hi.view(dt).double() + lo.view(dt).double()
    err = (rec - w.double()).abs()
    if npass == 2:
        bound = torch.maximum(w.double().abs() * 2.0 ** -21, torch.full_like(err, 6.0e-8))       # 2^-22 relative, or half a subnormal step
    else:
        bound = w.double().abs() * 2.0 ** -15
    assert (err <= bound).all(), (err / bound).max()
    if npass == 2:
        big = torch.tensor([[7.0e4, -9.9e4, 1.3e5, 3.0e38, float('nan'), float('inf'), float('-inf')] + [0.0] * 25])
        pl = ops.prepare_weight(big.to(dev), npass=2)
        rec = pl[0, :1].cpu().view(torch.float16).double() + pl[1, :1].cpu().view(torch.float16).double()
        # finite values beyond fp16's range saturate at +-65504 (a GEMM never manufactures an infinity from a finite operand); a NaN or
        # infinite PARAMETER comes out as a NaN lo half -- the weight preparation does not launder a diverged optimizer state
        assert torch.equal(rec[0, :4], torch.tensor([65504.0, -65504.0, 65504.0, 65504.0], dtype=torch.float64))
        assert torch.isnan(rec[0, 4:7]).all() and bool(torch.isfinite(rec[0, 7:]).all())


@pytest.mark.parametrize('npass', [2, 4])
@pytest.mark.parametrize('M,N,K', [(300, 256, 256), (128, 768, 256), (257, 192, 96), (90, 64, 64), (1000, 512, 256), (513, 131, 64), (2000, 256, 768)])
def test_gemm_nt_plain(dev, M, N, K, npass):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=npass)
    assert rel_err(out, ref) < TOL[npass]
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=npass, act=1, out_scale=2.5)
    assert rel_err(out, torch.relu(ref) * 2.5) < TOL[npass]
    if npass == 4 and _grad_hi_built():      # HFTT_NT_A_HI: A (a gradient in the backward) enters as its bf16 rounding, the weights keep their pair
        out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=4, grad_hi=True)
        assert rel_err(out, A.bfloat16().double() @ W.double().T + b.double()) < TOL[4]


def test_gemm_nt_small_and_large_magnitudes(dev):
    """fp16 halves: activations of ~1e3 against weights of ~1e-4 (lo halves are subnormals) still carry fp32-grade products."""
    ops = _ops()
    M, N, K = 256, 256, 256
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g) * 900.0
    W = torch.randn(N, K, generator=g) * 2e-4
    ref = A.double() @ W.double().T
    out = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=2)
    assert rel_err(out, ref) < 2e-4          # the weights' absolute floor (3e-8 per element: half an fp16 subnormal step) relative to 2e-4
    out3 = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=3)
    assert rel_err(out, out3.double()) < 2e-4
    # a flushed subnormal lo half would cost 6e-5 per weight, i.e. errors of order 0.1 here
    W2 = torch.randn(N, K, generator=g) * 0.05
    out = ops.gemm_nt(A.to(dev), W2.to(dev), None, npass=2)
    assert rel_err(out, A.double() @ W2.double().T) < 3e-6


@pytest.mark.parametrize('npass', [2, 4])
@pytest.mark.parametrize('N', [256, 64, 128])
def test_gemm_nt_epilogues(dev, N, npass):
    ops = _ops()
    M, K = 333, 128
    tol = TOL[npass]
    g = torch.Generator().manual_seed(N)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    table = torch.randn(7, N, generator=g); res = torch.randn(M, N, generator=g); res5 = torch.randn(5, N, generator=g)
    gate = torch.randn(M, N, generator=g); gam = torch.randn(N, generator=g); bet = torch.randn(N, generator=g)
    lin = A.double() @ W.double().T + b.double()
    rows = torch.arange(M)
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=npass, out_scale=3.0, add_table=table.to(dev), add_mod=7)
    assert rel_err(out, lin * 3.0 + table.double()[rows % 7]) < tol
    out = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=npass, gate=gate.to(dev), gate_scale=1.25)
    ref = torch.where(gate.double() > 0, (A.double() @ W.double().T) * 1.25, torch.zeros((), dtype=torch.float64))
    assert rel_err(out, ref) < tol
    p, site, seed = 0.3, 5, 77
    mask = keep_mask_t(seed, site, (M, N), p).double()
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=npass, drop_p=p, drop_site=site, drop_seed=seed, residual=res5.to(dev), res_mod=5)
    ref = lin * mask * keep_scale(p) + res5.double()[rows % 5]
    assert rel_err(out, ref) < tol
    out, pre, mean, rstd = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=npass, residual=res.to(dev), ln=(gam.to(dev), bet.to(dev)))
    r = lin + res.double()
    ref = F.layer_norm(r, (N,), gam.double(), bet.double(), 1e-5)
    assert rel_err(pre, r) < tol
    assert rel_err(out, ref) < 1e-4
    assert rel_err(mean, r.mean(1)) < 1e-4
    assert rel_err(rstd, 1.0 / torch.sqrt(r.var(1, unbiased=False) + 1e-5)) < 1e-4


@pytest.mark.parametrize('npass', [4, 2])
@pytest.mark.parametrize('M,N,K', [(5000, 256, 256), (1000, 192, 256), (777, 128, 96), (88, 64, 64), (4096, 768, 256), (3000, 512, 256), (2000, 256, 512)])
def test_gemm_tn(dev, M, N, K, npass):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g) * 1e-6          # gradient-sized: zero or subnormal as fp16, exact range as bf16
    X = torch.randn(M, K, generator=g)
    if npass == 2:
        dY = dY * 1e6
    dW, db = ops.gemm_tn(dY.to(dev), X.to(dev), npass=npass, out_scale=0.5)
    ref = 0.5 * dY.double().T @ X.double()
    scale = math.sqrt(M) * dY.abs().max().item()
    assert max_err(dW, ref) / scale < TOL[npass] * 3
    assert max_err(db, 0.5 * dY.double().sum(0)) / scale < 1e-5
    if npass == 4 and _grad_hi_built():      # HFTT_TN_DY_HI: dY enters the product as its bf16 rounding (the bias gradient still sums the fp32 values)
        dW, db = ops.gemm_tn(dY.to(dev), X.to(dev), npass=4, out_scale=0.5, grad_hi=True)
        assert max_err(dW, 0.5 * dY.bfloat16().double().T @ X.double()) / scale < TOL[4] * 3
        assert max_err(db, 0.5 * dY.double().sum(0)) / scale < 1e-5


def _attn_ref(q, k, v, H, mask=None, keep_sc=1.0):
    n, Lq, d = q.shape
    Lk = k.shape[1]
    dh = d // H
    qh = q.view(n, Lq, H, dh).transpose(1, 2); kh = k.view(n, Lk, H, dh).transpose(1, 2); vh = v.view(n, Lk, H, dh).transpose(1, 2)
    e = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    pr = torch.softmax(e, -1)
    pd = pr if mask is None else pr * mask * keep_sc
    o = (pd @ vh).transpose(1, 2).reshape(n, Lq, d)
    return o, pr, torch.logsumexp(e, -1)


GEOMS = [(5, 4, 256, 256, 64), (5, 4, 88, 256, 64), (5, 4, 88, 88, 64), (6, 4, 128, 128, 64),
         (4, 2, 48, 48, 32), (4, 2, 12, 48, 32), (3, 2, 256, 256, 32), (3, 2, 16, 16, 32), (3, 2, 12, 12, 32), (2, 1, 100, 70, 64)]


@pytest.mark.parametrize('qk_scale', [1.0, 40.0])
@pytest.mark.parametrize('n,H,Lq,Lk,dh', GEOMS)
def test_attention_fwd_bwd(dev, n, H, Lq, Lk, dh, qk_scale):
    """qk_scale 40: logits of ~1e4 with near one-hot rows (the reference's first encoder layer on raw log-mel input)."""
    ops = _ops()
    d = H * dh
    g = torch.Generator().manual_seed(Lq * 1000 + Lk + dh)
    if Lq == Lk:
        qkv = torch.randn(n, Lq, 3 * d, generator=g)
        qkv[..., :2 * d] *= qk_scale
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        dq_, dk_, dv_ = (qkv.to(dev)[..., i * d:(i + 1) * d] for i in range(3))
    else:
        q = torch.randn(n, Lq, d, generator=g) * qk_scale
        kv = torch.randn(n, Lk, 2 * d, generator=g)
        kv[..., :d] *= qk_scale
        k, v = kv[..., :d], kv[..., d:]
        dq_ = q.to(dev); kvd = kv.to(dev); dk_, dv_ = kvd[..., :d], kvd[..., d:]
    do = torch.randn(n, Lq, d, generator=g) * 1e-5        # gradient-sized
    q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    o_ref, p_ref, lse_ref = _attn_ref(q64, k64, v64, H)
    (o_ref * do.double()).sum().backward()
    out, lse, probs = ops.attn_fwd(dq_, dk_, dv_, H, npass=2, want_probs=True)
    # logits of magnitude L carry an fp32 rounding of ~L * 6e-8 in ANY fp32 implementation; a probability moves by that much (relative)
    lmax = (q64.detach().view(n, Lq, H, dh).transpose(1, 2) @ k64.detach().view(n, Lk, H, dh).transpose(1, 2).transpose(-1, -2)).abs().max().item() / math.sqrt(dh)
    ptol = 4e-6 + 4e-7 * lmax
    assert max_err(probs, p_ref) < ptol
    assert abs(probs.sum(-1).mean().item() - 1.0) < 1e-4
    assert rel_err(out, o_ref) < 2 * ptol
    assert max_err(lse[..., 0] / math.sqrt(dh) - torch.log(lse[..., 1]), lse_ref) < 1e-4 + 4e-7 * lmax      # lse[0]: the RAW row maximum (x3 / bf16 kernels)
    dq, dk, dv = ops.attn_bwd(dq_, dk_, dv_, out, lse, do.to(dev), H, npass=2)
    gtol = 2e-4 + 3 * ptol
    if qk_scale == 1.0:
        assert rel_err(dq, q64.grad) < gtol
        assert rel_err(dk, k64.grad) < gtol
    else:
        # rows that are one-hot to 1e-200 have gradients of 1e-265 in fp64: judge dq / dk against the scale of their factors instead
        nat = do.abs().max().item() * v.abs().max().item() * max(q.abs().max().item(), k.abs().max().item()) * math.sqrt(dh)
        assert max_err(dq, q64.grad) < gtol * nat
        assert max_err(dk, k64.grad) < gtol * nat
    assert rel_err(dv, v64.grad) < gtol


@pytest.mark.parametrize('npass,stored', [(1, 'bf16'), (1, 'f32'), (2, 'f32')])
@pytest.mark.parametrize('n,H,Lq,Lk,dh', [(4, 2, 48, 48, 32), (3, 4, 256, 256, 64), (3, 4, 88, 88, 64)])
def test_attention_with_scores_of_1e9(dev, n, H, Lq, Lk, dh, npass, stored):
    """Seen in training (the single-pass mode, 1,175 steps into the small configuration): raw scores of -4e9 .. -1e10 in a first-layer row.
    softmax is shift-invariant and the reference's fp32 `x - max` is exact for the maximum element, so such a row is still a clean
    (near one-hot) distribution there; a rounded `s*c - max*c` is off by ulp(1e9) = 64..128 in the EXPONENT and underflowed the whole
    row (sum = 0, 1/sum = inf, NaN from there on).  The inputs are small integers times 2^11, so every product and sum is exact in
    every mode and the fp64 reference can be matched closely, ties included."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    d = H * dh
    q = torch.randint(-8, 9, (n, Lq, d), generator=g).float() * 2048.0
    k = torch.randint(-8, 9, (n, Lk, d), generator=g).float() * 2048.0
    v = torch.randn(n, Lk, d, generator=g).bfloat16().float()
    do = torch.randn(n, Lq, d, generator=g)
    q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    o_ref, p_ref, _ = _attn_ref(q64, k64, v64, H)
    (o_ref * do.double()).sum().backward()
    assert (q64.detach().view(n, Lq, H, dh).transpose(1, 2) @ k64.detach().view(n, Lk, H, dh).transpose(1, 2).transpose(-1, -2)).amax(-1).abs().max().item() > 1e9
    dt = torch.bfloat16 if stored == 'bf16' else torch.float32            # bf16: the kernels of the bf16 activation stream (every value here is exact in it)
    qd, kd, vd = (t.to(dev).to(dt) for t in (q, k, v))
    out, lse, probs = ops.attn_fwd(qd, kd, vd, H, npass=npass, want_probs=True, out_dtype=dt)      # all three bf16: the bf16-stream kernels
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(lse).all()) and bool(torch.isfinite(probs).all())
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, out, lse, do.to(dev).to(dt), H, npass=npass, dq_dtype=dt, dkv_dtype=dt)
    for t in (dq, dk, dv):
        assert bool(torch.isfinite(t).all())
    assert max_err(probs.sum(-1), torch.ones(n, H, Lq, dtype=torch.float64)) < 1e-5
    if npass == 1 and stored == 'f32' and dh != 64:
        return      # this form folds 1/sqrt(dh) into q BEFORE the bf16 rounding (inexact unless a power of two): its own arithmetic, no exact ties
    out, dq, dk, dv = (t.float() for t in (out, dq, dk, dv))
    assert max_err(probs, p_ref) < 1e-5
    assert rel_err(out, o_ref) < (1e-5 if npass == 2 else 1e-2)
    assert rel_err(dv, v64.grad) < (1e-4 if npass == 2 else 2e-2)
    # tied maxima give O(1) entries of dS; everything else is 0: judged on the scale of the factors
    nat = do.abs().max().item() * v.abs().max().item() * 16384.0 * math.sqrt(dh)
    assert max_err(dq, q64.grad) < (2e-4 if npass == 2 else 3e-2) * nat
    assert max_err(dk, k64.grad) < (2e-4 if npass == 2 else 3e-2) * nat


def test_attention_shared_query_and_dropout(dev):
    """Layer-zero geometry: one query block shared by all sequences (seq stride 0) + dropout with the device RNG."""
    ops = _ops()
    n, H, Lq, Lk, dh = 4, 4, 88, 256, 64
    d = H * dh
    g = torch.Generator().manual_seed(3)
    q1 = torch.randn(1, Lq, d, generator=g); k = torch.randn(n, Lk, d, generator=g); v = torch.randn(n, Lk, d, generator=g)
    do = torch.randn(n, Lq, d, generator=g)
    p, site, seed = 0.1, 9, 12345
    mask = keep_mask_t(seed, site, (n, H, Lq, Lk), p).double()
    qd = q1.to(dev).expand(n, Lq, d)       # stride 0 over sequences
    q64 = q1.double().clone().requires_grad_(True); k64 = k.double().clone().requires_grad_(True); v64 = v.double().clone().requires_grad_(True)
    o_ref, p_ref, _ = _attn_ref(q64.expand(n, Lq, d), k64, v64, H, mask, keep_scale(p))
    (o_ref * do.double()).sum().backward()
    out, lse, probs = ops.attn_fwd(qd, k.to(dev), v.to(dev), H, npass=2, want_probs=True, drop_p=p, drop_site=site, drop_seed=seed)
    assert max_err(probs, p_ref) < 2e-5          # returned probabilities are PRE-dropout (model_spec2midi.py:360)
    assert rel_err(out, o_ref) < 1e-4
    dq, dk, dv = ops.attn_bwd(qd, k.to(dev), v.to(dev), out, lse, do.to(dev), H, npass=2, drop_p=p, drop_site=site, drop_seed=seed)
    assert rel_err(dq.sum(0, keepdim=True), q64.grad) < 3e-4
    assert rel_err(dk, k64.grad) < 3e-4
    assert rel_err(dv, v64.grad) < 3e-4


# ---------------------------------------------------------------------------------------------------------------------
# q / k / v as f16-pair planes (round 4): written once by the projection, staged by LDS-DMA in the forward (csrc/x3_attn_pl.hip)
# ---------------------------------------------------------------------------------------------------------------------
def _planes_ref(x):
    """the plane form of an fp32 tensor [..., cols] with torch's own fp16 rounding: per 32-column group 32 hi halves, then 32 lo halves"""
    g = x.reshape(-1, x.shape[-1] // 32, 32)
    hi = g.clamp(-65504.0, 65504.0).half()
    lo = (g - hi.float()).half()
    return torch.cat([hi, lo], -1).contiguous().view(torch.float32).reshape(x.shape)


def test_to_planes_and_the_projection_epilogue_write_the_same_bytes(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 256, generator=g) * torch.logspace(-5, 3, 256).unsqueeze(0)
    pl = ops.to_planes(x.to(dev))
    assert torch.equal(pl.cpu().view(torch.int32), _planes_ref(x).view(torch.int32))
    # hftt_strip_linear with HFTT_SL_C_F16PAIR == to_planes(the same launch's fp32 result), N = 256 / 512 / 768, ragged last block
    for M, N in ((384, 256), (1056, 768), (640, 512), (40000, 768)):
        xx = torch.randn(M, 256, generator=g); W = torch.randn(N, 256, generator=g) / 16.0; b = torch.randn(N, generator=g)
        wp = ops.x3_strip_pack(W.to(dev), 2, order=1)
        full = ops.strip_linear(xx.to(dev), wp, N, b.to(dev), x3=2)
        pln = ops.strip_linear(xx.to(dev), wp, N, b.to(dev), x3=2, c_planes=True)
        assert torch.equal(pln.view(torch.int32), ops.to_planes(full).view(torch.int32)), (M, N)
    # N = 1024 / 1536 (the cross-attention K / V projections of two / three decoder layers as ONE launch, round 6) exist as planes only: against
    # the fp32 results of the same rows of W taken 768 columns at a time (an output tile depends on its own rows of W only); 70,048 tokens =
    # 548 blocks for 512 resident workgroups
    for M, N in ((640, 1024), (1056, 1536), (70048, 1536)):
        xx = torch.randn(M, 256, generator=g); W = torch.randn(N, 256, generator=g) / 16.0; b = torch.randn(N, generator=g)
        pln = ops.strip_linear(xx.to(dev), ops.x3_strip_pack(W.to(dev), 2, order=1), N, b.to(dev), x3=2, c_planes=True)
        parts = []
        for n0 in range(0, N, 768):
            n1 = min(N, n0 + 768)
            parts.append(ops.strip_linear(xx.to(dev), ops.x3_strip_pack(W[n0:n1].contiguous().to(dev), 2, order=1), n1 - n0, b[n0:n1].contiguous().to(dev), x3=2))
        full = torch.cat(parts, 1).contiguous()
        assert torch.equal(pln.view(torch.int32), ops.to_planes(full).view(torch.int32)), (M, N)
        assert rel_err(full, xx.double() @ W.double().T + b.double()) < TOL[2]


PLANE_GEOMS = [(5, 4, 256, 256), (37, 4, 256, 256), (80, 4, 256, 256), (5, 4, 88, 256), (300, 4, 88, 256), (5, 4, 88, 88), (6, 4, 128, 128), (700, 2, 128, 128), (3, 2, 48, 48),
               (3, 1, 16, 16), (3, 2, 12, 40), (2, 1, 100, 70), (2, 2, 200, 250)]


@pytest.mark.parametrize('drop', [0.0, 0.1])
@pytest.mark.parametrize('n,H,Lq,Lk', PLANE_GEOMS)
def test_attention_on_planes_equals_attention_on_fp32_operands(dev, n, H, Lq, Lk, drop):
    """The plane forms run the SAME arithmetic in the same order on the same halves: forward outputs, row statistics and the attention map are
    bit-identical to the fp32-operand kernels' (themselves held against fp64 above); the backward's bf16 pairs are formed from hi + lo instead
    of the fp32 value (2^-23 apart: a bf16 pair's last bit, 2^-17, may round the other way), so its gradients agree to the pair's precision.  More (sequence, head) items than persistent workgroups -- in the forward AND in the backward (320 / 1,200 / 1,400 items on 256 / 512
    resident workgroups: second and third items per workgroup, unevenly) --, key padding,
    ragged query blocks, idle waves (88 queries on 4 waves), the shared query of layer zero."""
    ops = _ops()
    dh, d = 64, 64 * H
    g = torch.Generator().manual_seed(Lq * 1000 + Lk + n)
    if Lq == Lk:
        qkv = (torch.randn(n, Lq, 3 * d, generator=g) * 1.5).to(dev)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        pq = ops.to_planes(qkv)
        qp, kp, vp = pq[..., :d], pq[..., d:2 * d], pq[..., 2 * d:]
    else:
        q = (torch.randn(n, Lq, d, generator=g) * 1.5).to(dev)
        kv = (torch.randn(n, Lk, 2 * d, generator=g) * 1.5).to(dev)
        k, v = kv[..., :d], kv[..., d:]
        qp = ops.to_planes(q); pkv = ops.to_planes(kv)
        kp, vp = pkv[..., :d], pkv[..., d:]
    kw = dict(drop_p=drop, drop_site=7, drop_seed=99)
    for want in (False, True):
        ref = ops.attn_fwd(q, k, v, H, npass=2, want_probs=want, **kw)
        got = ops.attn_fwd(qp, kp, vp, H, npass=2, want_probs=want, planes=True, **kw)
        for a, b_ in zip(ref, got):
            assert torch.equal(a, b_), (want, max_err(a, b_))
    out, lse = ref[0], ref[1]
    do = (torch.randn(n, Lq, d, generator=g) * 1e-5).to(dev)
    r_dq, r_dk, r_dv = ops.attn_bwd(q, k, v, out, lse, do, H, npass=2, **kw)
    g_dq, g_dk, g_dv = ops.attn_bwd(qp, kp, vp, out, lse, do, H, npass=2, planes=True, **kw)
    for a, b_ in ((r_dq, g_dq), (r_dk, g_dk), (r_dv, g_dv)):
        assert rel_err(b_, a.double().cpu()) < 4e-5


def test_attention_on_planes_shared_query(dev):
    """layer zero (model_spec2midi.py:154-155): one query block for all sequences (sequence stride 0), as planes from hftt_x3_to_planes"""
    ops = _ops()
    n, H, Lq, Lk, d = 9, 4, 88, 256, 256
    g = torch.Generator().manual_seed(8)
    q1 = torch.randn(1, Lq, d, generator=g).to(dev); kv = torch.randn(n, Lk, 2 * d, generator=g).to(dev)
    qd = q1.expand(n, Lq, d); qp = ops.to_planes(q1).expand(n, Lq, d)
    pkv = ops.to_planes(kv)
    kw = dict(drop_p=0.1, drop_site=3, drop_seed=5)
    ref = ops.attn_fwd(qd, kv[..., :d], kv[..., d:], H, npass=2, want_probs=True, **kw)
    got = ops.attn_fwd(qp, pkv[..., :d], pkv[..., d:], H, npass=2, want_probs=True, planes=True, **kw)
    for a, b_ in zip(ref, got):
        assert torch.equal(a, b_)
    do = (torch.randn(n, Lq, d, generator=g) * 1e-5).to(dev)
    r = ops.attn_bwd(qd, kv[..., :d], kv[..., d:], ref[0], ref[1], do, H, npass=2, **kw)
    t = ops.attn_bwd(qp, pkv[..., :d], pkv[..., d:], ref[0], ref[1], do, H, npass=2, planes=True, **kw)
    for a, b_ in zip(r, t):
        assert rel_err(b_, a.double().cpu()) < 4e-5


# ---------------------------------------------------------------------------------------------------------------------
# strip kernels of the x3 mode (csrc/x3_strip.hip): token strips as (hi, lo) register pairs, split weight fragments through the ring
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('elem', [2, 4])
@pytest.mark.parametrize('M,N,K', [(384, 256, 256), (1056, 768, 256), (640, 512, 256), (416, 256, 512), (992, 256, 768), (40000, 768, 256),
                                   (70048, 256, 768), (70048, 256, 512), (70048, 768, 256)])
def test_strip_linear(dev, M, N, K, elem):
    """(M = 70,048 = 548 blocks of 128 tokens, the last one ragged: more blocks than the 512 workgroups the two-per-CU forms keep resident, so the
    path that hands a workgroup its NEXT block -- strip chunks loaded across the block boundary -- runs under a reference too, not only in the bench.)"""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    wp = ops.x3_strip_pack(W.to(dev), elem, order=1 if K == 256 else 0)      # K == 256 without LayerNorm: the tile-major kernel
    ref = x.double() @ W.double().T + b.double()
    out = ops.strip_linear(x.to(dev), wp, N, b.to(dev), x3=elem)
    assert rel_err(out, ref) < TOL[elem]
    out = ops.strip_linear(x.to(dev), wp, N, b.to(dev), relu=True, out_scale=0.5, residual=res.to(dev), x3=elem)
    assert rel_err(out, torch.relu(ref) * 0.5 + res.double()) < TOL[elem]
    p, site, seed = 0.1, 3, 4242
    mask = keep_mask_t(seed, site, (M, N), p).double()
    out = ops.strip_linear(x.to(dev), wp, N, b.to(dev), drop_p=p, drop_site=site, drop_seed=seed, residual=res[:7].contiguous().to(dev), res_mod=7, x3=elem)
    assert rel_err(out, ref * mask * keep_scale(p) + res.double()[torch.arange(M) % 7]) < TOL[elem]
    if elem == 4 and _grad_hi_built():       # HFTT_SL_X3_GRAD_HI: the strip (a gradient in the backward) enters as its bf16 rounding, the weights keep their pair
        out = ops.strip_linear(x.to(dev), wp, N, b.to(dev), residual=res.to(dev), x3=4, grad_hi=True)
        assert rel_err(out, x.bfloat16().double() @ W.double().T + b.double() + res.double()) < TOL[4]


@pytest.mark.parametrize('elem', [2, 4])
@pytest.mark.parametrize('K', [256, 512])
@pytest.mark.parametrize('M', [1120, 40032])       # 40,032 tokens = 313 blocks: more than the 256 resident workgroups (next-block strip prefetch)
def test_strip_linear_layernorm(dev, K, elem, M):
    ops = _ops()
    N = 256
    g = torch.Generator().manual_seed(K)
    x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) * 3.0; gam = torch.randn(N, generator=g); bet = torch.randn(N, generator=g)
    p, site, seed = 0.1, 11, 99
    mask = keep_mask_t(seed, site, (M, N), p).double()
    wp = ops.x3_strip_pack(W.to(dev), elem)
    out, pre, mean, rstd = ops.strip_linear(x.to(dev), wp, N, b.to(dev), drop_p=p, drop_site=site, drop_seed=seed, residual=res.to(dev),
                                            ln=(gam.to(dev), bet.to(dev)), x3=elem)
    r = (x.double() @ W.double().T + b.double()) * mask * keep_scale(p) + res.double()
    assert rel_err(pre, r) < TOL[elem]
    assert rel_err(out, F.layer_norm(r, (N,), gam.double(), bet.double(), 1e-5)) < 1e-4
    assert rel_err(mean, r.mean(1)) < 1e-4
    assert rel_err(rstd, 1.0 / torch.sqrt(r.var(1, unbiased=False) + 1e-5)) < 1e-4
    out2 = ops.strip_linear(x.to(dev), wp, N, b.to(dev), drop_p=p, drop_site=site, drop_seed=seed, residual=res.to(dev),
                            ln=(gam.to(dev), bet.to(dev)), save_pre=False, x3=elem)
    assert max_err(out2[0], out) == 0.0
    # the saved pre-LayerNorm sum as bf16 (HFTT_SL_PRE_BF16: what the x3 strip plans keep for the LayerNorm backward); the output is the same
    out3, pre3, mean3, rstd3 = ops.strip_linear(x.to(dev), wp, N, b.to(dev), drop_p=p, drop_site=site, drop_seed=seed, residual=res.to(dev),
                                                ln=(gam.to(dev), bet.to(dev)), x3=elem, pre_bf16=True)
    assert pre3.dtype == torch.bfloat16 and rel_err(pre3.float(), r) < 4e-3
    assert max_err(out3, out) == 0.0 and max_err(mean3, mean) == 0.0 and max_err(rstd3, rstd) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# The three consumers of a LayerNorm backward's result that mask it with the dropout site WHILE THEY LOAD IT (round 6: the LayerNorm backward
# writes no masked copy): HFTT_SL_X_DROP (fc_o dX), HFTT_TN_DY_DROP (weight-gradient product), site_o of hftt_ffn_bwd_dx.  Each against fp64
# on the masked input, with more token blocks than resident workgroups and -- for the product -- a site of 2^26 elements (32-bit quad index).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M', [992, 70048])
def test_strip_linear_masks_a_dropout_gradient_on_load(dev, M):
    ops = _ops()
    N = K = 256
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K)
    p, site, seed = 0.1, 21, 777
    xm = x.double() * keep_mask_t(seed, site, (M, K), p).double() * keep_scale(p)
    wp = ops.x3_strip_pack(W.to(dev), 4, order=1)
    out = ops.strip_linear(x.to(dev), wp, N, None, drop_p=p, drop_site=site, drop_seed=seed, x3=4, x_drop=True)
    assert rel_err(out, xm @ W.double().T) < TOL[4]
    wrong = ops.strip_linear(x.to(dev), wp, N, None, drop_p=p, drop_site=site + 1, drop_seed=seed, x3=4, x_drop=True)
    assert rel_err(wrong, xm @ W.double().T) > 0.1            # (the neighbouring site is a different mask)


@pytest.mark.parametrize('x_bf', [False, True])
@pytest.mark.parametrize('M,N,K', [(5000, 256, 256), (5000, 256, 512), (262144, 256, 256)])
def test_gemm_tn_masks_a_dropout_gradient_on_load(dev, M, N, K, x_bf):
    ops = _ops()
    g = torch.Generator().manual_seed(M + K)
    dY = torch.randn(M, N, generator=g); X = torch.randn(M, K, generator=g)
    if x_bf:
        X = X.bfloat16()
    p, site, seed = 0.1, 5, 31337
    dYm = dY.double() * keep_mask_t(seed, site, (M, N), p).double() * keep_scale(p)
    dW, db = ops.gemm_tn(dY.to(dev), X.to(dev), npass=4, out_scale=0.5, dy_drop=(p, site, seed))
    assert rel_err(dW, 0.5 * dYm.T @ X.double()) < TOL[4]
    assert rel_err(db, 0.5 * dYm.sum(0)) < 3e-5
    dW0, db0 = ops.gemm_tn(dY.to(dev), X.to(dev), npass=4, out_scale=0.5, dy_drop=(0.0, site, seed))      # p = 0: the unmasked product
    assert rel_err(dW0, 0.5 * dY.double().T @ X.double()) < TOL[4] and rel_err(db0, 0.5 * dY.double().sum(0)) < 3e-5


@pytest.mark.parametrize('M', [992, 33024])
def test_fused_ffn_dx_masks_the_output_dropout_gradient_on_load(dev, M):
    ops = _ops()
    d, pf = 256, 512
    g = torch.Generator().manual_seed(M + 1)
    dy = torch.randn(M, d, generator=g); hid = torch.randn(M, pf, generator=g)
    W1 = torch.randn(pf, d, generator=g) / math.sqrt(d); W2 = torch.randn(d, pf, generator=g) / math.sqrt(pf)
    res = torch.randn(M, d, generator=g)
    p, site, seed = 0.1, 9, 4711
    wb = ops.x3_ffn_pack(W1.to(dev), W2.to(dev), backward=True)
    dym = dy.double() * keep_mask_t(seed, site, (M, d), p).double() * keep_scale(p)
    dh_ref = (hid.double() > 0) * (dym @ W2.double()) * 1.25
    dx, dh = ops.ffn_bwd_dx(dy.to(dev), wb, pf, hid.to(dev), gate_scale=1.25, residual=res.to(dev), x3=True, dy_drop=(p, site, seed))
    assert rel_err(dh, dh_ref) < TOL[4]
    assert rel_err(dx, dh_ref @ W1.double() + res.double()) < TOL[4]


@pytest.mark.parametrize('side', ['dY', 'X'])
@pytest.mark.parametrize('M,N,K', [(5000, 256, 512), (5000, 512, 256), (1000, 192, 256), (777, 128, 96), (88, 64, 64), (3000, 256, 256)])
def test_gemm_tn_with_one_operand_stored_as_bf16(dev, M, N, K, side):
    """npass 4 with the FFN hidden (X of dW2) or its gradient (dY of dW1) stored as bf16: that operand is its own hi half, the other one is
    still split; the product of the VALUES STORED is reproduced to the split-bf16 tolerance."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g) * 1e-6
    X = torch.randn(M, K, generator=g)
    if side == 'dY':
        dY = dY.bfloat16()
    else:
        X = X.bfloat16()
    dW, db = ops.gemm_tn(dY.to(dev), X.to(dev), npass=4, out_scale=0.5)
    ref = 0.5 * dY.double().T @ X.double()
    scale = math.sqrt(M) * dY.float().abs().max().item()
    assert max_err(dW, ref) / scale < TOL[4] * 3
    assert max_err(db, 0.5 * dY.double().sum(0)) / scale < 1e-5
    if side == 'X' and _grad_hi_built():     # + HFTT_TN_DY_HI: the plain bf16 product of the stored hidden and the rounded gradient (what the engine's dW2 is)
        dW, db = ops.gemm_tn(dY.to(dev), X.to(dev), npass=4, out_scale=0.5, grad_hi=True)
        assert max_err(dW, 0.5 * dY.bfloat16().double().T @ X.double()) / scale < TOL[4] * 3


@pytest.mark.parametrize('hbf', [False, True])
@pytest.mark.parametrize('M', [256, 4000 * 32 // 32 * 1 + 96, 33024])
def test_fused_ffn_forward_and_dx(dev, M, hbf):
    ops = _ops()
    d, pf = 256, 512
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, d, generator=g); W1 = torch.randn(pf, d, generator=g) / 16.0; W2 = torch.randn(d, pf, generator=g) / 22.0
    b1 = torch.randn(pf, generator=g) * 0.3; b2 = torch.randn(d, generator=g) * 0.3; gam = torch.randn(d, generator=g); bet = torch.randn(d, generator=g)
    p, sh, so, seed = 0.1, 21, 22, 777
    wf = ops.x3_ffn_pack(W1.to(dev), W2.to(dev))
    y, hid, pre, mean, rstd = ops.ffn_res_ln_fwd(x.to(dev), wf, pf, b1.to(dev), b2.to(dev), gam.to(dev), bet.to(dev), drop_p=p, site_h=sh, site_o=so, seed=seed, x3=True,
                                                 hidden_bf16=hbf, pre_bf16=hbf)
    h = torch.relu(x.double() @ W1.double().T + b1.double()) * keep_mask_t(seed, sh, (M, pf), p).double() * keep_scale(p)
    o = (h @ W2.double().T + b2.double()) * keep_mask_t(seed, so, (M, d), p).double() * keep_scale(p)
    r = x.double() + o
    assert hid.dtype == (torch.bfloat16 if hbf else torch.float32)
    assert rel_err(hid.float(), h) < (4e-3 if hbf else 4e-6)          # (the STORED copy; fc_2 took the hidden from registers: `pre` below)
    assert rel_err(pre.float(), r) < (4e-3 if hbf else 4e-6)          # (the copy saved for the LayerNorm backward; y below is normalised from registers)
    assert rel_err(y, F.layer_norm(r, (d,), gam.double(), bet.double(), 1e-5)) < 1e-4
    assert rel_err(mean, r.mean(1)) < 1e-4
    # inference form: nothing saved, same result
    y2 = ops.ffn_res_ln_fwd(x.to(dev), wf, pf, b1.to(dev), b2.to(dev), gam.to(dev), bet.to(dev), drop_p=p, site_h=sh, site_o=so, seed=seed,
                            save_hidden=False, save_pre=False, x3=True)[0]
    assert max_err(y2, y) == 0.0
    # dX half of the backward (gradient-sized dy)
    dy = torch.randn(M, d, generator=g) * 1e-5; res = torch.randn(M, d, generator=g) * 1e-5
    wb = ops.x3_ffn_pack(W1.to(dev), W2.to(dev), backward=True)
    dx, dh = ops.ffn_bwd_dx(dy.to(dev), wb, pf, hid, gate_scale=1.25, residual=res.to(dev), x3=True)
    # (the gate is the DEVICE's stored hidden: a pre-activation within rounding of zero may fall on either side of the ReLU)
    dh_ref = torch.where(hid.cpu().double() > 0, (dy.double() @ W2.double()) * 1.25, torch.zeros((), dtype=torch.float64))
    assert dh.dtype == hid.dtype
    assert rel_err(dh.float(), dh_ref) < (4e-3 if hbf else 6e-5)       # (stored copy; dx is formed from the full-width dh in registers)
    assert rel_err(dx, dh_ref @ W1.double() + res.double()) < 6e-5
    if not _grad_hi_built():
        return
    # HFTT_SL_X3_GRAD_HI: dy and the dh formed from it enter their products as bf16 roundings
    dx2, dh2 = ops.ffn_bwd_dx(dy.to(dev), wb, pf, hid, gate_scale=1.25, residual=res.to(dev), x3=True, grad_hi=True)
    dh_r = torch.where(hid.cpu().double() > 0, (dy.bfloat16().double() @ W2.double()) * 1.25, torch.zeros((), dtype=torch.float64))
    assert rel_err(dh2.float(), dh_r) < (4e-3 if hbf else 6e-5)
    assert rel_err(dx2, dh_r.float().bfloat16().double() @ W1.double() + res.double()) < 1.5e-3      # (dh is rounded from the device's fp32 value, not from this fp64 one: ties fall either way)


# ---------------------------------------------------------------------------------------------------------------------
# the small-width family (csrc/x3s_strip.h): the reference's default model, d = 64 / ff = 128 (training/m_training.py:56-61)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('elem', [2, 4])
@pytest.mark.parametrize('M,N,K', [(384, 64, 64), (1056, 192, 64), (640, 128, 64), (416, 64, 128), (992, 64, 192), (100000, 192, 64), (33024, 64, 192)])
def test_small_strip_linear(dev, M, N, K, elem):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    wp = ops.x3s_pack(W.to(dev), elem)
    ref = x.double() @ W.double().T + b.double()
    if K == 64:
        out = ops.strip_linear(x.to(dev), wp, N, b.to(dev), x3=elem)
        assert rel_err(out, ref) < TOL[elem]
    if N == 64:
        out = ops.strip_linear(x.to(dev), wp, N, b.to(dev), relu=True, out_scale=0.5, residual=res.to(dev), x3=elem)
        assert rel_err(out, torch.relu(ref) * 0.5 + res.double()) < TOL[elem]
        p, site, seed = 0.1, 3, 4242
        mask = keep_mask_t(seed, site, (M, N), p).double()
        out = ops.strip_linear(x.to(dev), wp, N, b.to(dev), drop_p=p, drop_site=site, drop_seed=seed, residual=res[:7].contiguous().to(dev), res_mod=7, x3=elem)
        assert rel_err(out, ref * mask * keep_scale(p) + res.double()[torch.arange(M) % 7]) < TOL[elem]
    # a transposed pack (the dX kernels): Wl = W^T
    if elem == 4 and K == 64:
        dy = torch.randn(M, N, generator=g) * 1e-5
        wt = ops.x3s_pack(W.to(dev), 4, transpose=True)            # logical [K, N]: maps [M, N] -> [M, K]
        if N in (64, 128, 192) and K == 64:
            out = ops.strip_linear(dy.to(dev), wt, K, None, x3=4)
            assert rel_err(out, dy.double() @ W.double()) < TOL[4]


@pytest.mark.parametrize('elem', [2, 4])
def test_small_strip_linear_layernorm(dev, elem):
    ops = _ops()
    M, N, K = 1120, 64, 64
    g = torch.Generator().manual_seed(K)
    x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) * 3.0; gam = torch.randn(N, generator=g); bet = torch.randn(N, generator=g)
    p, site, seed = 0.1, 11, 99
    mask = keep_mask_t(seed, site, (M, N), p).double()
    wp = ops.x3s_pack(W.to(dev), elem)
    out, pre, mean, rstd = ops.strip_linear(x.to(dev), wp, N, b.to(dev), drop_p=p, drop_site=site, drop_seed=seed, residual=res.to(dev),
                                            ln=(gam.to(dev), bet.to(dev)), x3=elem)
    r = (x.double() @ W.double().T + b.double()) * mask * keep_scale(p) + res.double()
    assert rel_err(pre, r) < TOL[elem]
    assert rel_err(out, F.layer_norm(r, (N,), gam.double(), bet.double(), 1e-5)) < 1e-4
    assert rel_err(mean, r.mean(1)) < 1e-4
    assert rel_err(rstd, 1.0 / torch.sqrt(r.var(1, unbiased=False) + 1e-5)) < 1e-4
    out3, pre3, mean3, rstd3 = ops.strip_linear(x.to(dev), wp, N, b.to(dev), drop_p=p, drop_site=site, drop_seed=seed, residual=res.to(dev),
                                                ln=(gam.to(dev), bet.to(dev)), x3=elem, pre_bf16=True)
    assert pre3.dtype == torch.bfloat16 and rel_err(pre3.float(), r) < 4e-3
    assert max_err(out3, out) == 0.0 and max_err(mean3, mean) == 0.0 and max_err(rstd3, rstd) == 0.0


@pytest.mark.parametrize('hbf', [False, True])
@pytest.mark.parametrize('M,res_mod', [(256, 0), (4096 + 96, 0), (33024, 0), (1056, 88)])
def test_attention_output_and_ffn_as_one_launch_equal_the_two_launches_bit_for_bit(dev, M, res_mod, hbf):
    """hftt_attn_out_ffn_fwd (ABI v8) == hftt_strip_linear (fc_o + dropout + residual + LayerNorm) followed by hftt_ffn_res_ln_fwd: every stored
    tensor bit for bit, in the training form (everything a backward needs is written) and in the inference form (x1 is never written); the two
    launches themselves are held against fp64 by test_strip_linear_layernorm and test_fused_ffn_forward_and_dx.  33,024 tokens = 258 blocks for
    256 resident workgroups; res_mod: the broadcast residual of the decoder's layer zero."""
    ops = _ops()
    d, pf = 256, 512
    g = torch.Generator().manual_seed(M + 7)
    ctx = torch.randn(M, d, generator=g); Wo = torch.randn(d, d, generator=g) / 16.0; bo = torch.randn(d, generator=g) * 0.3
    res = torch.randn(res_mod or M, d, generator=g)
    W1 = torch.randn(pf, d, generator=g) / 16.0; W2 = torch.randn(d, pf, generator=g) / 22.0
    b1 = torch.randn(pf, generator=g) * 0.3; b2 = torch.randn(d, generator=g) * 0.3
    g1, be1, g2, be2 = (torch.randn(d, generator=g) for _ in range(4))
    p, sa, sh, so, seed = 0.1, 20, 21, 22, 777
    D = lambda t: t.to(dev)
    x1, pre1, m1, r1 = ops.strip_linear(D(ctx), ops.x3_strip_pack(D(Wo), 2), d, D(bo), drop_p=p, drop_site=sa, drop_seed=seed, residual=D(res), res_mod=res_mod,
                                        ln=(D(g1), D(be1)), x3=2, pre_bf16=hbf)
    y, hid, pre2, m2, r2 = ops.ffn_res_ln_fwd(x1, ops.x3_ffn_pack(D(W1), D(W2)), pf, D(b1), D(b2), D(g2), D(be2), drop_p=p, site_h=sh, site_o=so, seed=seed, x3=True,
                                              hidden_bf16=hbf, pre_bf16=hbf)
    wp = ops.x3_attn_out_ffn_pack(D(Wo), D(W1), D(W2))
    got = ops.attn_out_ffn_fwd(D(ctx), wp, D(bo), D(res), D(g1), D(be1), pf, D(b1), D(b2), D(g2), D(be2), drop_p=p, site_a=sa, site_h=sh, site_o=so, seed=seed,
                               res_mod=res_mod, hidden_bf16=hbf, pre_bf16=hbf)
    for name, a, b_ in zip(('y', 'x1', 'pre1', 'mean1', 'rstd1', 'hidden', 'pre2', 'mean2', 'rstd2'), got, (y, x1, pre1, m1, r1, hid, pre2, m2, r2)):
        assert a.dtype == b_.dtype and torch.equal(a, b_), name
    y_inf = ops.attn_out_ffn_fwd(D(ctx), wp, D(bo), D(res), D(g1), D(be1), pf, D(b1), D(b2), D(g2), D(be2), drop_p=p, site_a=sa, site_h=sh, site_o=so, seed=seed,
                                 res_mod=res_mod, save=False)
    assert torch.equal(y_inf, y)


@pytest.mark.parametrize('hbf', [False, True])
@pytest.mark.parametrize('M', [256, 4096 + 96, 90112])
def test_small_fused_ffn_forward_and_dx(dev, M, hbf):
    ops = _ops()
    d, pf = 64, 128
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, d, generator=g); W1 = torch.randn(pf, d, generator=g) / 8.0; W2 = torch.randn(d, pf, generator=g) / 11.0
    b1 = torch.randn(pf, generator=g) * 0.3; b2 = torch.randn(d, generator=g) * 0.3; gam = torch.randn(d, generator=g); bet = torch.randn(d, generator=g)
    p, sh, so, seed = 0.1, 21, 22, 777
    wf = ops.x3s_ffn_pack(W1.to(dev), W2.to(dev))
    y, hid, pre, mean, rstd = ops.ffn_res_ln_fwd(x.to(dev), wf, pf, b1.to(dev), b2.to(dev), gam.to(dev), bet.to(dev), drop_p=p, site_h=sh, site_o=so, seed=seed, x3=True,
                                                 hidden_bf16=hbf, pre_bf16=hbf)
    h = torch.relu(x.double() @ W1.double().T + b1.double()) * keep_mask_t(seed, sh, (M, pf), p).double() * keep_scale(p)
    o = (h @ W2.double().T + b2.double()) * keep_mask_t(seed, so, (M, d), p).double() * keep_scale(p)
    r = x.double() + o
    assert hid.dtype == (torch.bfloat16 if hbf else torch.float32)
    assert rel_err(hid.float(), h) < (4e-3 if hbf else 4e-6)
    assert rel_err(pre.float(), r) < (4e-3 if hbf else 4e-6)
    assert rel_err(y, F.layer_norm(r, (d,), gam.double(), bet.double(), 1e-5)) < 1e-4
    assert rel_err(mean, r.mean(1)) < 1e-4
    y2 = ops.ffn_res_ln_fwd(x.to(dev), wf, pf, b1.to(dev), b2.to(dev), gam.to(dev), bet.to(dev), drop_p=p, site_h=sh, site_o=so, seed=seed,
                            save_hidden=False, save_pre=False, x3=True)[0]
    assert max_err(y2, y) == 0.0
    dy = torch.randn(M, d, generator=g) * 1e-5; res = torch.randn(M, d, generator=g) * 1e-5
    wb = ops.x3s_ffn_pack(W1.to(dev), W2.to(dev), backward=True)
    dx, dh = ops.ffn_bwd_dx(dy.to(dev), wb, pf, hid, gate_scale=1.25, residual=res.to(dev), x3=True)
    dh_ref = torch.where(hid.cpu().double() > 0, (dy.double() @ W2.double()) * 1.25, torch.zeros((), dtype=torch.float64))
    assert dh.dtype == hid.dtype
    assert rel_err(dh.float(), dh_ref) < (4e-3 if hbf else 6e-5)
    assert rel_err(dx, dh_ref @ W1.double() + res.double()) < 6e-5
