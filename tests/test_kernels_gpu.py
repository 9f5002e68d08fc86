"""Per-kernel parity of the HIP kernels (through the C ABI) against fp64 torch references on seeded inputs."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util
from util import keep_scale, rel_err, max_err, keep_mask_t

pytestmark = pytest.mark.gpu

TOL = {3: 3e-6, 1: 2e-2}     # relative to the output's max magnitude: exact-fp32 parity mode / single-pass bf16


def _ops():
    from hftt_hip import ops
    return ops


@pytest.mark.parametrize('npass', [3, 1])
@pytest.mark.parametrize('M,N,K', [(300, 256, 256), (128, 768, 256), (257, 192, 96), (90, 64, 64), (1000, 512, 256), (513, 131, 64)])
def test_gemm_nt_plain(dev, M, N, K, npass):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=npass)
    assert rel_err(out, ref) < TOL[npass]
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=npass, act=1, out_scale=2.5)
    assert rel_err(out, torch.relu(ref) * 2.5) < TOL[npass]


@pytest.mark.parametrize('N', [256, 64, 128])
def test_gemm_nt_epilogues(dev, N):
    ops = _ops()
    M, K = 333, 128
    g = torch.Generator().manual_seed(N)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    table = torch.randn(7, N, generator=g); res = torch.randn(M, N, generator=g); res5 = torch.randn(5, N, generator=g)
    gate = torch.randn(M, N, generator=g); gam = torch.randn(N, generator=g); bet = torch.randn(N, generator=g)
    lin = A.double() @ W.double().T + b.double()
    rows = torch.arange(M)
    # scale + table add (position embedding)
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), out_scale=3.0, add_table=table.to(dev), add_mod=7)
    assert rel_err(out, lin * 3.0 + table.double()[rows % 7]) < TOL[3]
    # gate (ReLU / dropout backward)
    out = ops.gemm_nt(A.to(dev), W.to(dev), None, gate=gate.to(dev), gate_scale=1.25)
    ref = torch.where(gate.double() > 0, (A.double() @ W.double().T) * 1.25, torch.zeros((), dtype=torch.float64))
    assert rel_err(out, ref) < TOL[3]
    # dropout then residual (post-norm residual branch), broadcast residual with res_mod
    p, site, seed = 0.3, 5, 77
    mask = keep_mask_t(seed, site, (M, N), p).double()
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), drop_p=p, drop_site=site, drop_seed=seed, residual=res5.to(dev), res_mod=5)
    ref = lin * mask * keep_scale(p) + res5.double()[rows % 5]
    assert rel_err(out, ref) < TOL[3]
    # residual + LayerNorm
    out, pre, mean, rstd = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), residual=res.to(dev), ln=(gam.to(dev), bet.to(dev)))
    r = lin + res.double()
    ref = F.layer_norm(r, (N,), gam.double(), bet.double(), 1e-5)
    assert rel_err(pre, r) < TOL[3]
    assert rel_err(out, ref) < 1e-4
    assert rel_err(mean, r.mean(1)) < 1e-4
    assert rel_err(rstd, 1.0 / torch.sqrt(r.var(1, unbiased=False) + 1e-5)) < 1e-4


@pytest.mark.parametrize('npass', [3, 1])
@pytest.mark.parametrize('M,N,K', [(5000, 256, 256), (1000, 192, 256), (777, 128, 96), (88, 64, 64), (4096, 768, 256), (3000, 512, 256), (2000, 256, 512)])
def test_gemm_tn(dev, M, N, K, npass):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g); X = torch.randn(M, K, generator=g)
    dW, db = ops.gemm_tn(dY.to(dev), X.to(dev), npass=npass, out_scale=0.5)
    ref = 0.5 * dY.double().T @ X.double()
    scale = math.sqrt(M)
    assert max_err(dW, ref) / scale < TOL[npass] * 3
    assert max_err(db, 0.5 * dY.double().sum(0)) / scale < 1e-5


def _attn_ref(q, k, v, H, mask=None, keep=1.0):
    n, Lq, d = q.shape
    Lk = k.shape[1]
    dh = d // H
    qh = q.view(n, Lq, H, dh).transpose(1, 2); kh = k.view(n, Lk, H, dh).transpose(1, 2); vh = v.view(n, Lk, H, dh).transpose(1, 2)
    e = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    pr = torch.softmax(e, -1)
    pd = pr if mask is None else pr * mask / keep
    o = (pd @ vh).transpose(1, 2).reshape(n, Lq, d)
    return o, pr, torch.logsumexp(e, -1)


GEOMS = [(5, 4, 256, 256, 64), (5, 4, 88, 256, 64), (5, 4, 88, 88, 64), (6, 4, 128, 128, 64),
         (4, 2, 48, 48, 32), (4, 2, 12, 48, 32), (3, 2, 256, 256, 32), (3, 2, 16, 16, 32), (3, 2, 12, 12, 32), (2, 1, 100, 70, 64)]


@pytest.mark.parametrize('npass', [3, 1])
@pytest.mark.parametrize('n,H,Lq,Lk,dh', GEOMS)
def test_attention_fwd_bwd(dev, n, H, Lq, Lk, dh, npass):
    ops = _ops()
    d = H * dh
    g = torch.Generator().manual_seed(Lq * 1000 + Lk + dh)
    # strided layouts as the engine uses them: q/k/v are column blocks of a fused [n, L, 3d] projection when Lq == Lk
    if Lq == Lk:
        qkv = torch.randn(n, Lq, 3 * d, generator=g)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        dq_, dk_, dv_ = (qkv.to(dev)[..., i * d:(i + 1) * d] for i in range(3))
    else:
        q = torch.randn(n, Lq, d, generator=g)
        kv = torch.randn(n, Lk, 2 * d, generator=g)
        k, v = kv[..., :d], kv[..., d:]
        dq_ = q.to(dev); kvd = kv.to(dev); dk_, dv_ = kvd[..., :d], kvd[..., d:]
    do = torch.randn(n, Lq, d, generator=g)
    q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    o_ref, p_ref, lse_ref = _attn_ref(q64, k64, v64, H)
    (o_ref * do.double()).sum().backward()
    out, lse, probs = ops.attn_fwd(dq_, dk_, dv_, H, npass=npass, want_probs=True)
    tol = TOL[npass]
    assert max_err(probs, p_ref) < (2e-6 if npass == 3 else 2e-2)
    assert abs(probs.sum(-1).mean().item() - 1.0) < 1e-4
    assert rel_err(out, o_ref) < tol * 2
    # (fp32-stored operands: lse[0] is the maximum of the SCALED scores in both modes; the all-bf16 / x3 kernels keep the raw one, hftt_hip.h)
    assert max_err(lse[..., 0] - torch.log(lse[..., 1]), lse_ref) < (1e-4 if npass == 3 else 5e-2)
    dq, dk, dv = ops.attn_bwd(dq_, dk_, dv_, out, lse, do.to(dev), H, npass=npass)
    assert rel_err(dq, q64.grad) < tol * 4
    assert rel_err(dk, k64.grad) < tol * 4
    assert rel_err(dv, v64.grad) < tol * 4


def test_attention_shared_query_and_dropout(dev):
    """Layer-zero geometry: one query block shared by all sequences (seq stride 0) + dropout with the device RNG."""
    ops = _ops()
    n, H, Lq, Lk, dh = 4, 4, 88, 256, 64
    d = H * dh
    g = torch.Generator().manual_seed(3)
    q1 = torch.randn(1, Lq, d, generator=g); k = torch.randn(n, Lk, d, generator=g); v = torch.randn(n, Lk, d, generator=g)
    do = torch.randn(n, Lq, d, generator=g)
    p, site, seed = 0.25, 9, 12345
    mask = keep_mask_t(seed, site, (n, H, Lq, Lk), p).double()
    keep = 1.0 - float(np.float32(p))
    qd = q1.to(dev).expand(n, Lq, d)       # stride 0 over sequences
    q64 = q1.double().clone().requires_grad_(True); k64 = k.double().clone().requires_grad_(True); v64 = v.double().clone().requires_grad_(True)
    o_ref, p_ref, _ = _attn_ref(q64.expand(n, Lq, d), k64, v64, H, mask, keep)
    (o_ref * do.double()).sum().backward()
    out, lse, probs = ops.attn_fwd(qd, k.to(dev), v.to(dev), H, want_probs=True, drop_p=p, drop_site=site, drop_seed=seed)
    assert max_err(probs, p_ref) < 2e-5          # returned probabilities are PRE-dropout (model_spec2midi.py:360)
    assert rel_err(out, o_ref) < 1e-4
    dq, dk, dv = ops.attn_bwd(qd, k.to(dev), v.to(dev), out, lse, do.to(dev), H, drop_p=p, drop_site=site, drop_seed=seed)
    assert rel_err(dq.sum(0, keepdim=True), q64.grad) < 2e-4
    assert rel_err(dk, k64.grad) < 2e-4
    assert rel_err(dv, v64.grad) < 2e-4


@pytest.mark.parametrize('p', [0.0, 0.1])
@pytest.mark.parametrize('n,H,Lq,Lk,dh', [(4, 4, 16, 16, 64), (4, 2, 48, 48, 32), (5, 4, 256, 256, 64), (5, 4, 88, 256, 64), (4, 4, 128, 128, 64), (3, 2, 12, 48, 32)])
def test_bf16_stream_attention_with_dropout(dev, n, H, Lq, Lk, dh, p):
    """The kernels of the single-pass mode as the engine drives them: q / k / v, the context and the gradients all STORED as bf16, dropout
    from the device RNG, against fp64 on the same (bf16-valued) inputs and the same mask."""
    ops = _ops()
    d = H * dh
    g = torch.Generator().manual_seed(17)
    bf = torch.bfloat16
    q = torch.randn(n, Lq, d, generator=g).to(bf); k = torch.randn(n, Lk, d, generator=g).to(bf); v = torch.randn(n, Lk, d, generator=g).to(bf)
    do = torch.randn(n, Lq, d, generator=g).to(bf)
    site, seed = 5, 777
    mask = keep_mask_t(seed, site, (n, H, Lq, Lk), p).double() if p > 0 else None
    q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    o_ref, p_ref, _ = _attn_ref(q64, k64, v64, H, mask, 1.0 / keep_scale(p))
    (o_ref * do.double()).sum().backward()
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    out, lse, probs = ops.attn_fwd(qd, kd, vd, H, npass=1, want_probs=True, drop_p=p, drop_site=site, drop_seed=seed, out_dtype=bf)
    assert max_err(probs, p_ref) < 2e-2
    assert rel_err(out.float(), o_ref) < 2e-2
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, out, lse, do.to(dev), H, npass=1, drop_p=p, drop_site=site, drop_seed=seed, dq_dtype=bf, dkv_dtype=bf)
    assert rel_err(dq.float(), q64.grad) < 3e-2
    assert rel_err(dk.float(), k64.grad) < 3e-2
    assert rel_err(dv.float(), v64.grad) < 3e-2


@pytest.mark.parametrize('p', [0.0, 0.1])
@pytest.mark.parametrize('n,H,Lq,Lk,dh', [(64, 4, 16, 16, 64), (16, 4, 128, 128, 64), (16, 2, 48, 48, 32), (8, 4, 88, 256, 64)])
def test_bf16_stream_attention_backward_with_nearly_identical_keys(dev, n, H, Lq, Lk, dh, p):
    """The decoder's self-attention over time at initialisation: the keys of a sequence are one common vector plus a 1 % spread.  dQ = dS.K
    with sum_j dS_ij = 0 is then a small difference of large terms; with dS rounded to bf16 the single-pass kernel returned cos(dQ, fp64) of
    0.1 .. 0.4 (and the mode stalled in training with dropout on) until the mean key was taken off the dQ operand (csrc/attn_bwd.hip)."""
    ops = _ops()
    bf = torch.bfloat16
    d = H * dh
    g = torch.Generator().manual_seed(5)
    q = torch.randn(n, Lq, d, generator=g).to(bf)
    k = (torch.randn(n, 1, d, generator=g) * 3 + 0.02 * torch.randn(n, Lk, d, generator=g)).to(bf)
    v = (torch.randn(n, 1, d, generator=g) * 3 + torch.randn(n, Lk, d, generator=g)).to(bf)
    do = torch.randn(n, Lq, d, generator=g).to(bf)
    site, seed = 5, 777
    mask = keep_mask_t(seed, site, (n, H, Lq, Lk), p).double() if p > 0 else None
    q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    o_ref, _, _ = _attn_ref(q64, k64, v64, H, mask, 1.0 / keep_scale(p))
    (o_ref * do.double()).sum().backward()
    a = [t.to(dev) for t in (q, k, v)]
    out, lse = ops.attn_fwd(*a, H, npass=1, drop_p=p, drop_site=site, drop_seed=seed, out_dtype=bf)
    dq, dk, dv = ops.attn_bwd(*a, out, lse, do.to(dev), H, npass=1, drop_p=p, drop_site=site, drop_seed=seed, dq_dtype=bf, dkv_dtype=bf)

    def cos(x, y):
        x = x.double().cpu().flatten(); y = y.flatten()
        return float(x @ y / (x.norm() * y.norm()))
    assert cos(dq, q64.grad) > 0.995 and cos(dk, k64.grad) > 0.995 and cos(dv, v64.grad) > 0.995


@pytest.mark.parametrize('N,M', [(256, 1234), (256, 4099), (256, 3), (128, 1234), (64, 1234)])
def test_ln_bwd(dev, N, M):
    ops = _ops()
    g = torch.Generator().manual_seed(N)
    r = torch.randn(M, N, generator=g) * 2 + 0.5; dy = torch.randn(M, N, generator=g); gam = torch.randn(N, generator=g)
    r64 = r.double().requires_grad_(True); g64 = gam.double().requires_grad_(True); b64 = torch.zeros(N, dtype=torch.float64, requires_grad=True)
    y = F.layer_norm(r64, (N,), g64, b64, 1e-5)
    (y * dy.double()).sum().backward()
    mean = r.double().mean(1); rstd = 1.0 / torch.sqrt(r.double().var(1, unbiased=False) + 1e-5)
    p, site, seed = 0.2, 3, 99
    dr, drd, dg, db = ops.ln_bwd(dy.to(dev), r.to(dev), mean.float().to(dev), rstd.float().to(dev), gam.to(dev), drop_p=p, drop_site=site, drop_seed=seed)
    assert rel_err(dr, r64.grad) < 1e-5
    assert rel_err(dg, g64.grad) < 1e-5
    assert rel_err(db, b64.grad) < 1e-5
    mask = keep_mask_t(seed, site, (M, N), p).double()
    assert rel_err(drd, r64.grad * mask * keep_scale(p)) < 1e-5


@pytest.mark.parametrize('M,p', [(1234, 0.2), (4099, 0.0), (3, 0.1), (65536, 0.1)])
def test_ln_bwd_all_bf16_four_rows_per_wave(dev, M, p):
    """N = 256 with dy, r, dr and the dropped copy stored as bf16 (the strip plans' gradient stream): the four-rows-per-wave kernel, ragged M
    included, against the fp64 LayerNorm backward of the SAME bf16-rounded operands."""
    ops = _ops()
    N = 256
    g = torch.Generator().manual_seed(M)
    r = (torch.randn(M, N, generator=g) * 2 + 0.5).to(torch.bfloat16); dy = torch.randn(M, N, generator=g).to(torch.bfloat16)
    gam = torch.randn(N, generator=g)
    r64 = r.double().requires_grad_(True); g64 = gam.double().requires_grad_(True); b64 = torch.zeros(N, dtype=torch.float64, requires_grad=True)
    (F.layer_norm(r64, (N,), g64, b64, 1e-5) * dy.double()).sum().backward()
    mean = r.double().mean(1); rstd = 1.0 / torch.sqrt(r.double().var(1, unbiased=False) + 1e-5)
    site, seed = 5, 1234
    dr, drd, dg, db = ops.ln_bwd(dy.to(dev), r.to(dev), mean.float().to(dev), rstd.float().to(dev), gam.to(dev), drop_p=p, drop_site=site,
                                 drop_seed=seed, drop_dtype=torch.bfloat16, dr_dtype=torch.bfloat16)
    assert dr.dtype == torch.bfloat16
    assert rel_err(dr, r64.grad) < 6e-3                 # one bf16 rounding of the output
    assert rel_err(dg, g64.grad) < 2e-5 and rel_err(db, b64.grad) < 2e-5
    if p > 0:
        mask = keep_mask_t(seed, site, (M, N), p).double()
        assert drd.dtype == torch.bfloat16 and rel_err(drd, r64.grad * mask * keep_scale(p)) < 6e-3
        assert torch.equal((drd == 0).cpu() | (mask == 1), torch.ones(M, N, dtype=torch.bool))      # exactly the masked elements are zero


@pytest.mark.parametrize('M,p,bf', [(1234, 0.2, False), (4099, 0.0, False), (3, 0.1, False), (17, 0.1, True), (65536, 0.1, True)])
def test_ln_bwd_width_64_four_rows_per_wave(dev, M, p, bf):
    """N = 64 (the reference's default width): the four-rows-per-wave kernel, ragged M included (M % 4 != 0, M < 16), fp32 and bf16 storage,
    against the fp64 LayerNorm backward of the same (rounded) operands."""
    ops = _ops()
    N = 64
    g = torch.Generator().manual_seed(M)
    r = torch.randn(M, N, generator=g) * 2 + 0.5; dy = torch.randn(M, N, generator=g); gam = torch.randn(N, generator=g)
    if bf:
        r, dy = r.to(torch.bfloat16), dy.to(torch.bfloat16)
    r64 = r.double().requires_grad_(True); g64 = gam.double().requires_grad_(True); b64 = torch.zeros(N, dtype=torch.float64, requires_grad=True)
    (F.layer_norm(r64, (N,), g64, b64, 1e-5) * dy.double()).sum().backward()
    mean = r.double().mean(1); rstd = 1.0 / torch.sqrt(r.double().var(1, unbiased=False) + 1e-5)
    site, seed = 5, 1234
    kw = dict(drop_dtype=torch.bfloat16, dr_dtype=torch.bfloat16) if bf else {}
    dr, drd, dg, db = ops.ln_bwd(dy.to(dev), r.to(dev), mean.float().to(dev), rstd.float().to(dev), gam.to(dev), drop_p=p, drop_site=site, drop_seed=seed, **kw)
    tol = 6e-3 if bf else 1e-5
    assert rel_err(dr, r64.grad) < tol
    assert rel_err(dg, g64.grad) < 2e-5 and rel_err(db, b64.grad) < 2e-5
    if p > 0:
        mask = keep_mask_t(seed, site, (M, N), p).double()
        assert rel_err(drd, r64.grad * mask * keep_scale(p)) < tol
        assert torch.equal((drd == 0).cpu() | (mask == 1), torch.ones(M, N, dtype=torch.bool))


def test_colsum_and_adam(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 3000, generator=g)
    out = ops.colsum(x.to(dev))
    assert max_err(out, x.double().sum(0)) < 1e-3
    out2 = ops.colsum(x.to(dev), beta=1.0, out=out.clone())
    assert max_err(out2, 2 * x.double().sum(0)) < 2e-3
    n = 100003
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g) * 0.1
    m = torch.zeros(n); v = torch.zeros(n)
    pd, md, vd = p.to(dev).clone(), m.to(dev), v.to(dev)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    for step in (1, 2, 3):
        ops.adam_step(pd, gr.to(dev), md, vd, step, lr=1e-3)
        util.O.adam_step([pr], [gr], [mr], [vr], step, lr=1e-3)
    assert max_err(pd, pr) < 1e-6
    assert max_err(vd, vr) < 1e-8


def test_logmel(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    n = 16000 * 2 + 123
    t = torch.arange(n) / 16000.0
    wave = 0.3 * torch.sin(2 * math.pi * 220.0 * t) * torch.exp(-2.0 * t) + 0.1 * torch.sin(2 * math.pi * 1760.0 * t) + 0.01 * torch.randn(n, generator=g)
    lm = ops.LogMel(dev)
    feat = lm(wave.to(dev))
    ref = util.O.logmel_dft(wave)
    assert feat.shape == ref.shape == (1 + n // 256, 256)
    assert max_err(feat, ref) < 2e-3
    ref32 = util.O.logmel(wave)
    assert max_err(feat, ref32) < 5e-3


# ------------------------------------------------------------------------------------------------------------------
# bf16-STORED operands (io_flags): numerically the same kernels, inputs are exactly representable in bf16 here so the
# only extra error is the final rounding of a bf16 output (2^-9 relative)
# ------------------------------------------------------------------------------------------------------------------
def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize('M,N,K', [(1000, 256, 256), (700, 768, 256), (515, 256, 512), (300, 256, 768), (257, 192, 96), (4096, 512, 256)])
def test_gemm_nt_bf16_storage(dev, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    gate = _bf(torch.randn(M, N, generator=g)); res = torch.randn(M, N, generator=g)
    lin = A.double() @ _bf(W).double().T + b.double()
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=1)                                    # bf16 A, fp32 C
    assert out.dtype == torch.float32 and rel_err(out, lin) < 1e-5
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=1, act=1, out_dtype=torch.bfloat16)    # bf16 A, bf16 C
    assert out.dtype == torch.bfloat16 and rel_err(out.float(), torch.relu(lin)) < 5e-3
    out = ops.gemm_nt(A.float().to(dev), W.to(dev), None, npass=1, gate=gate.to(dev), gate_scale=2.0, residual=res.to(dev), out_dtype=torch.bfloat16)
    ref = torch.where(gate.double() > 0, (lin - b.double()) * 2.0, torch.zeros((), dtype=torch.float64)) + res.double()
    assert rel_err(out.float(), ref) < 5e-3
    # elementwise extras on a bf16 C (ReLU + dropout; bf16 ReLU gate): the direct packed-store epilogue when N % 256 == 0 and K <= 256
    out = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=1, act=1, drop_p=0.25, drop_site=7, drop_seed=99, out_dtype=torch.bfloat16)
    keep = util.keep_mask_t(99, 7, (M, N), 0.25)
    ref = torch.where(keep, torch.relu(lin) / 0.75, torch.zeros((), dtype=torch.float64))
    assert rel_err(out.float(), ref) < 5e-3 and ((out.float().cpu() == 0) >= ~keep).all()
    out = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=1, gate=gate.to(dev), gate_scale=1.5, out_dtype=torch.bfloat16)
    ref = torch.where(gate.double() > 0, (lin - b.double()) * 1.5, torch.zeros((), dtype=torch.float64))
    assert rel_err(out.float(), ref) < 5e-3 and ((out.float().cpu() == 0) >= (gate <= 0)).all()
    if N == 256:
        gam = torch.randn(N, generator=g); bet = torch.randn(N, generator=g)
        o, pre, mean, rstd = ops.gemm_nt(A.to(dev), W.to(dev), b.to(dev), npass=1, residual=res.to(dev), ln=(gam.to(dev), bet.to(dev)))
        assert rel_err(pre, lin + res.double()) < 1e-5
        assert rel_err(o, F.layer_norm(lin + res.double(), (N,), gam.double(), bet.double(), 1e-5)) < 1e-4


# (N, K <= 256 take the 128 x 256 tile, the wider shapes the 256 x 256 tile; the two long cases are the model's S_n and S_e row counts)
@pytest.mark.parametrize('M,N,K', [(5000, 256, 256), (3000, 512, 256), (1000, 192, 256), (2000, 768, 256), (90112, 256, 256), (262144, 256, 256)])
def test_gemm_tn_bf16_storage(dev, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    dY = _bf(torch.randn(M, N, generator=g)); X = _bf(torch.randn(M, K, generator=g))
    ref = dY.double().T @ X.double()
    for a, b in ((dY, X), (dY.float(), X), (dY, X.float())):
        dW, db = ops.gemm_tn(a.to(dev), b.to(dev), npass=1)
        assert max_err(dW, ref) / math.sqrt(M) < 1e-4
        assert max_err(db, dY.double().sum(0)) / math.sqrt(M) < 1e-5


@pytest.mark.parametrize('n,H,Lq,Lk,dh', [(5, 4, 256, 256, 64), (5, 4, 88, 256, 64), (4, 4, 88, 88, 64), (4, 4, 128, 128, 64), (3, 2, 48, 48, 32)])
def test_attention_bf16_storage(dev, n, H, Lq, Lk, dh):
    ops = _ops()
    d = H * dh
    g = torch.Generator().manual_seed(Lq + Lk)
    q = _bf(torch.randn(n, Lq, d, generator=g)); k = _bf(torch.randn(n, Lk, d, generator=g)); v = _bf(torch.randn(n, Lk, d, generator=g))
    do = _bf(torch.randn(n, Lq, d, generator=g))
    q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    o_ref, p_ref, _ = _attn_ref(q64, k64, v64, H)
    out, lse, probs = ops.attn_fwd(q.to(dev), k.to(dev), v.to(dev), H, npass=1, want_probs=True, out_dtype=torch.bfloat16)
    assert out.dtype == torch.bfloat16
    assert max_err(probs, p_ref) < 2e-2 and rel_err(out.float(), o_ref) < 3e-2
    # backward against autograd THROUGH the bf16-rounded output the kernel itself stored
    (o_ref * do.double()).sum().backward()
    dq, dk, dv = ops.attn_bwd(q.to(dev), k.to(dev), v.to(dev), out, lse, do.to(dev), H, npass=1, dq_dtype=torch.bfloat16, dkv_dtype=torch.bfloat16)
    assert dq.dtype == dk.dtype == torch.bfloat16
    assert rel_err(dq.float(), q64.grad) < 6e-2 and rel_err(dk.float(), k64.grad) < 6e-2 and rel_err(dv.float(), v64.grad) < 6e-2
    # same kernels with fp32 storage agree with the bf16-storage run up to the output roundings
    out32, lse32 = ops.attn_fwd(q.float().to(dev), k.float().to(dev), v.float().to(dev), H, npass=1)
    assert rel_err(out.float(), out32) < 1e-2


def test_bf16_gradient_stream_flags(dev):
    """bf16 gradient stream (ABI: HFTT_LNB_DY_BF16 / HFTT_LNB_DR_BF16, HFTT_NT_RES_BF16): with values that are exactly
    representable in bf16 the bf16-stored inputs give bitwise the fp32-stored result; a bf16 output differs by its rounding only."""
    ops = _ops()
    M, N, K = 1500, 256, 256
    g = torch.Generator().manual_seed(12)
    r = torch.randn(M, N, generator=g); dy = _bf(torch.randn(M, N, generator=g)); gam = torch.randn(N, generator=g)
    mean = r.mean(1); rstd = 1.0 / torch.sqrt(r.var(1, unbiased=False) + 1e-5)
    args = (r.to(dev), mean.to(dev), rstd.to(dev), gam.to(dev))
    a = ops.ln_bwd(dy.float().to(dev), *args, drop_p=0.2, drop_site=3, drop_seed=9, drop_dtype=torch.bfloat16)
    b = ops.ln_bwd(dy.to(dev), *args, drop_p=0.2, drop_site=3, drop_seed=9, drop_dtype=torch.bfloat16)                       # bf16 dy
    c = ops.ln_bwd(dy.to(dev), *args, drop_p=0.2, drop_site=3, drop_seed=9, drop_dtype=torch.bfloat16, dr_dtype=torch.bfloat16)  # + bf16 dr
    assert b[0].dtype == torch.float32 and max_err(a[0], b[0]) == 0.0 and max_err(a[2], b[2]) == 0.0 and max_err(a[3], b[3]) == 0.0
    assert torch.equal(a[1], b[1]) and torch.equal(a[1], c[1])
    assert c[0].dtype == torch.bfloat16 and torch.equal(c[0], a[0].to(torch.bfloat16))
    # NT with a bf16-stored residual, fp32 and bf16 C (row-pass epilogue), and under the LayerNorm epilogue
    A = _bf(torch.randn(M, K, generator=g)); W = torch.randn(N, K, generator=g) / math.sqrt(K); res = _bf(torch.randn(M, N, generator=g))
    x0 = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=1, residual=res.float().to(dev))
    x1 = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=1, residual=res.to(dev))
    x2 = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=1, residual=res.to(dev), out_dtype=torch.bfloat16)
    assert x1.dtype == torch.float32 and torch.equal(x0, x1) and torch.equal(x2, x0.to(torch.bfloat16))
    ref = A.double() @ _bf(W).double().T + res.double()
    assert rel_err(x1, ref) < 1e-5
    gam2 = torch.randn(N, generator=g); bet2 = torch.randn(N, generator=g)
    l0 = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=1, residual=res.float().to(dev), ln=(gam2.to(dev), bet2.to(dev)))
    l1 = ops.gemm_nt(A.to(dev), W.to(dev), None, npass=1, residual=res.to(dev), ln=(gam2.to(dev), bet2.to(dev)))
    assert all(torch.equal(p, q) for p, q in zip(l0, l1))
    # K = 512 / 768 shapes take other instantiations of the same kernel
    for K2 in (512, 768):
        A2 = _bf(torch.randn(M, K2, generator=g)); W2 = torch.randn(N, K2, generator=g) / math.sqrt(K2)
        y0 = ops.gemm_nt(A2.to(dev), W2.to(dev), None, npass=1, residual=res.float().to(dev))
        y1 = ops.gemm_nt(A2.to(dev), W2.to(dev), None, npass=1, residual=res.to(dev), out_dtype=torch.bfloat16)
        assert torch.equal(y1, y0.to(torch.bfloat16))
    # a bf16 residual on a shape that falls back to the k-tiled kernel is refused, not mis-read
    with pytest.raises(RuntimeError):
        ops.gemm_nt(A[:100].to(dev), W.to(dev), None, npass=1, residual=res[:100].to(dev))


def test_ln_bwd_bf16_dropped_output(dev):
    ops = _ops()
    M, N = 777, 256
    g = torch.Generator().manual_seed(4)
    r = torch.randn(M, N, generator=g); dy = torch.randn(M, N, generator=g); gam = torch.randn(N, generator=g)
    mean = r.mean(1); rstd = 1.0 / torch.sqrt(r.var(1, unbiased=False) + 1e-5)
    a = ops.ln_bwd(dy.to(dev), r.to(dev), mean.to(dev), rstd.to(dev), gam.to(dev), drop_p=0.2, drop_site=3, drop_seed=9)
    b = ops.ln_bwd(dy.to(dev), r.to(dev), mean.to(dev), rstd.to(dev), gam.to(dev), drop_p=0.2, drop_site=3, drop_seed=9, drop_dtype=torch.bfloat16)
    assert b[1].dtype == torch.bfloat16
    assert max_err(a[0], b[0]) == 0.0
    assert rel_err(b[1].float(), a[1]) < 5e-3
