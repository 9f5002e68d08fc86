"""Development experiment (test infrastructure, CPU only): which arithmetic meets north_star's 1e-3?

Emulates, on the CPU oracle at paper size, the operand roundings a device precision mode would apply, and reports the
max-abs error of the eight outputs against the fp32 oracle and against an fp64 evaluation of the same graph.

    python tests/dev_precision_emul.py [variant ...]

Variants (operand arithmetic of every linear / matmul; accumulation is fp32 as on the MFMA):
  bf16      one pass, operands rounded to bf16
  bf16x3    a = hi + lo (bf16 each): hi.hi + hi.lo + lo.hi
  fp16x3    the same with fp16 halves
  +res16    additionally round every tensor that crosses a kernel boundary (residual stream, q/k/v, context, hidden) to bf16
  +l0       ... but keep encoder layer 0's logits path exact (embedding, x0, Q/K projection, QK^T in fp32)
"""
import math
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from tests import util as U           # noqa: E402
from oracle import hftt_oracle as O   # noqa: E402


class Arith:
    def __init__(self, kind, res16=False, l0_exact=False, store='bf16'):
        self.kind, self.res16, self.l0_exact, self.store = kind, res16, l0_exact, store
        self.exact_now = False

    def split(self, x, dt):
        hi = x.to(dt).float()
        lo = (x - hi).to(dt).float()
        return hi, lo

    def mm(self, a, b):
        """a [.., m, k] @ b [.., k, n] with the mode's operand arithmetic"""
        if self.kind == 'fp32' or self.exact_now:
            return a @ b
        if self.kind == 'bf16':
            return a.bfloat16().float() @ b.bfloat16().float()
        if self.kind == 'fp16':
            return a.half().float() @ b.half().float()
        dt = torch.bfloat16 if self.kind == 'bf16x3' else torch.float16
        ah, al = self.split(a, dt)
        bh, bl = self.split(b, dt)
        return ah @ bh + (ah @ bl + al @ bh)

    def linear(self, x, w, b):
        y = self.mm(x, w.t())
        return y + b if b is not None else y

    def st(self, x):
        """a tensor that crosses a kernel boundary"""
        if self.res16 and not self.exact_now:
            return x.bfloat16().float() if self.store == 'bf16' else x.half().float()
        return x


def mha(A, sd, pre, q_in, k_in, v_in, H, exact_qk=False):
    bsz, lq, d = q_in.shape
    lk = k_in.shape[1]
    dh = d // H
    A.exact_now = exact_qk
    q = A.st(A.linear(q_in, sd[pre + 'fc_q.weight'], sd[pre + 'fc_q.bias']))
    k = A.st(A.linear(k_in, sd[pre + 'fc_k.weight'], sd[pre + 'fc_k.bias']))
    A.exact_now = False
    v = A.st(A.linear(v_in, sd[pre + 'fc_v.weight'], sd[pre + 'fc_v.bias']))
    q = q.view(bsz, lq, H, dh).transpose(1, 2)
    k = k.view(bsz, lk, H, dh).transpose(1, 2)
    v = v.view(bsz, lk, H, dh).transpose(1, 2)
    A.exact_now = exact_qk
    energy = A.mm(q, k.transpose(-1, -2)) / math.sqrt(dh)
    A.exact_now = False
    prob = torch.softmax(energy, dim=-1)
    ctx = A.st(A.mm(prob, v)).transpose(1, 2).contiguous().view(bsz, lq, d)
    return A.linear(ctx, sd[pre + 'fc_o.weight'], sd[pre + 'fc_o.bias']), prob


def ln(sd, pre, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + 'layer_norm.weight'], sd[pre + 'layer_norm.bias'], 1e-5)


def ffn(A, sd, pre, x):
    h = torch.relu(A.linear(x, sd[pre + 'fc_1.weight'], sd[pre + 'fc_1.bias']))      # (fused kernel: hidden stays on chip, rounded as an operand only)
    return A.linear(h, sd[pre + 'fc_2.weight'], sd[pre + 'fc_2.bias'])


def enc_layer(A, sd, pre, x, H, exact_qk=False, x_exact=None):
    """x_exact: the un-rounded layer input for the exact Q/K path of layer 0"""
    xin = x_exact if (exact_qk and x_exact is not None) else x
    if exact_qk:
        # q, k from the exact input; v and the residual from the stored one
        a, _ = mha_mixed(A, sd, pre + 'self_attention.', xin, x, H)
    else:
        a, _ = mha(A, sd, pre + 'self_attention.', x, x, x, H)
    x = A.st(ln(sd, pre, x + a))
    f = ffn(A, sd, pre + 'positionwise_feedforward.', x)
    return A.st(ln(sd, pre, x + f))


def mha_mixed(A, sd, pre, x_exact, x_st, H):
    bsz, lq, d = x_st.shape
    dh = d // H
    A.exact_now = True
    q = F.linear(x_exact, sd[pre + 'fc_q.weight'], sd[pre + 'fc_q.bias'])
    k = F.linear(x_exact, sd[pre + 'fc_k.weight'], sd[pre + 'fc_k.bias'])
    A.exact_now = False
    v = A.st(A.linear(x_st, sd[pre + 'fc_v.weight'], sd[pre + 'fc_v.bias']))
    q = q.view(bsz, lq, H, dh).transpose(1, 2); k = k.view(bsz, lq, H, dh).transpose(1, 2); v = v.view(bsz, lq, H, dh).transpose(1, 2)
    energy = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    prob = torch.softmax(energy, dim=-1)
    ctx = A.st(A.mm(prob, v)).transpose(1, 2).contiguous().view(bsz, lq, d)
    return A.linear(ctx, sd[pre + 'fc_o.weight'], sd[pre + 'fc_o.bias']), prob


def forward(A, sd, spec, cfg):
    pre = 'encoder_spec2midi.'
    bsz = spec.shape[0]
    T, Fq, N, V, d = cfg.n_frame, cfg.n_bin, cfg.n_note, cfg.n_velocity, cfg.hid_dim
    win = spec.unfold(2, cfg.n_proc, 1).permute(0, 2, 1, 3).contiguous().reshape(bsz * T, 1, Fq, cfg.n_proc)
    # the device folds conv + flatten + Linear(244, d) into one Linear(65 -> d): emulate the operand rounding on the folded form
    cw, cb = sd[pre + 'conv.weight'], sd[pre + 'conv.bias']              # [C,1,1,k], [C]
    tw, tb = sd[pre + 'tok_embedding_freq.weight'], sd[pre + 'tok_embedding_freq.bias']   # [d, C*nw]
    C_, kk = cw.shape[0], cw.shape[3]
    nw = cfg.n_proc - (kk - 1)
    tw3 = tw.view(d, C_, nw).double()
    weff = torch.zeros(d, cfg.n_proc, dtype=torch.float64)
    for c in range(C_):
        for t in range(kk):
            weff[:, t:t + nw] += tw3[:, c, :] * cw[c, 0, 0, t].double()
    beff = tb.double() + (tw3 * cb.double().view(1, C_, 1)).sum((1, 2))
    weff, beff = weff.to(spec.dtype), beff.to(spec.dtype)
    w2 = win.reshape(bsz * T, Fq, cfg.n_proc)
    pos = sd[pre + 'pos_embedding_freq.weight'][:Fq]
    A.exact_now = False
    x_exact = F.linear(w2, weff, beff) * math.sqrt(d) + pos.unsqueeze(0)
    x = A.st(A.linear(w2, weff, beff) * math.sqrt(d) + pos.unsqueeze(0)) if not A.l0_exact else A.st(x_exact)
    for i in range(cfg.enc_layer):
        x = enc_layer(A, sd, f'{pre}layers_freq.{i}.', x, cfg.enc_head, exact_qk=(A.l0_exact and i == 0), x_exact=x_exact)
    enc = x
    pre = 'decoder_spec2midi.'
    q0 = sd[pre + 'pos_embedding_freq.weight'][:N].unsqueeze(0).expand(bsz * T, N, d)
    p0 = pre + 'layer_zero_freq.'
    a, prob = mha(A, sd, p0 + 'encoder_attention.', q0, enc, enc, cfg.dec_head)
    trg = A.st(ln(sd, p0, q0 + a))
    trg = A.st(ln(sd, p0, trg + ffn(A, sd, p0 + 'positionwise_feedforward.', trg)))
    for i in range(cfg.dec_layer - 1):
        pl = f'{pre}layers_freq.{i}.'
        a, _ = mha(A, sd, pl + 'self_attention.', trg, trg, trg, cfg.dec_head)
        trg = A.st(ln(sd, pl, trg + a))
        a, prob = mha(A, sd, pl + 'encoder_attention.', trg, enc, enc, cfg.dec_head)
        trg = A.st(ln(sd, pl, trg + a))
        trg = A.st(ln(sd, pl, trg + ffn(A, sd, pl + 'positionwise_feedforward.', trg)))

    def heads(z, tag):
        o = [torch.sigmoid(A.linear(z, sd[f'{pre}fc_{nm}_{tag}.weight'], sd[f'{pre}fc_{nm}_{tag}.bias']).squeeze(-1)) for nm in ('onset', 'offset', 'mpe')]
        return o + [A.linear(z, sd[f'{pre}fc_velocity_{tag}.weight'], sd[f'{pre}fc_velocity_{tag}.bias'])]
    ha = heads(trg, 'freq')
    y = trg.reshape(bsz, T, N, d).permute(0, 2, 1, 3).contiguous().reshape(bsz * N, T, d)
    y = A.st(y * math.sqrt(d) + sd[pre + 'pos_embedding_time.weight'][:T].unsqueeze(0))
    for i in range(cfg.dec_layer):
        y = enc_layer(A, sd, f'{pre}layers_time.{i}.', y, cfg.dec_head)
    hb = heads(y, 'time')
    hb = [t.reshape(bsz, N, T).permute(0, 2, 1) for t in hb[:3]] + [hb[3].reshape(bsz, N, T, V).permute(0, 2, 1, 3)]
    ha = [t.reshape(bsz, T, N) for t in ha[:3]] + [ha[3].reshape(bsz, T, N, V)]
    return ha + [prob] + hb


VARIANTS = {
    'fp32': dict(kind='fp32'),
    'bf16': dict(kind='bf16'),
    'bf16+res16': dict(kind='bf16', res16=True),
    'bf16+res16+l0': dict(kind='bf16', res16=True, l0_exact=True),
    'bf16+l0': dict(kind='bf16', l0_exact=True),
    'fp16+l0': dict(kind='fp16', l0_exact=True),
    'fp16+resfp16+l0': dict(kind='fp16', res16=True, l0_exact=True, store='fp16'),
    'bf16x3': dict(kind='bf16x3'),
    'fp16x3': dict(kind='fp16x3'),
    'bf16x3+l0': dict(kind='bf16x3', l0_exact=True),
    'bf16x3+res_fp16x2': dict(kind='bf16x3'),
}


def main():
    names = sys.argv[1:] or list(VARIANTS)
    cfg = O.PAPER
    torch.set_num_threads(8)
    model = U.build_model(cfg, 1234)
    U.perturb(model, 99)
    sd = U.sd_cpu(model)
    spec = O.synth_spec(1, cfg, salt=5)
    with torch.no_grad():
        t0 = time.time()
        ref = forward(Arith('fp32'), sd, spec, cfg)
        print('fp32 forward %.1f s' % (time.time() - t0), flush=True)
        o_ref = O.model_forward(sd, spec, cfg)
        print('restated graph vs oracle: probs %.2e logits %.2e' % (max(U.max_err(ref[i], o_ref[i]) for i in (0, 1, 2, 5, 6, 7)),
                                                                    max(U.max_err(ref[i], o_ref[i]) for i in (3, 8))), flush=True)
        sd64 = {k: v.double() for k, v in sd.items()}
        r64 = forward(Arith('fp32'), sd64, spec.double(), cfg)
        print('fp32 vs fp64: probs A %.2e B %.2e | logits A %.2e B %.2e' % (
            max(U.max_err(ref[i], r64[i]) for i in (0, 1, 2)), max(U.max_err(ref[i], r64[i]) for i in (5, 6, 7)),
            U.max_err(ref[3], r64[3]), U.max_err(ref[8], r64[8])), flush=True)
        for nm in names:
            if nm == 'fp32':
                continue
            out = forward(Arith(**VARIANTS[nm]), sd, spec, cfg)
            flips = ((out[7] >= 0.5) != (ref[7] >= 0.5)).float().mean().item()
            print('%-18s vs fp32: probs A %.2e B %.2e | logits A %.2e B %.2e | attn %.2e | mpe_B flips %.4f   (vs fp64: probs %.2e logits %.2e)' % (
                nm, max(U.max_err(out[i], ref[i]) for i in (0, 1, 2)), max(U.max_err(out[i], ref[i]) for i in (5, 6, 7)),
                U.max_err(out[3], ref[3]), U.max_err(out[8], ref[8]), U.max_err(out[4], ref[4]), flips,
                max(U.max_err(out[i], r64[i]) for i in (0, 1, 2, 5, 6, 7)), max(U.max_err(out[i], r64[i]) for i in (3, 8))), flush=True)


if __name__ == '__main__':
    main()
