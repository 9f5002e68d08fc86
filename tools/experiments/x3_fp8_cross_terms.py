#!/usr/bin/env python3
"""Numerics of the proposal in DESIGN section 8 (1a): the x3 product  x.W = x_hi.W_hi + x_hi.W_lo + x_lo.W_hi  with the two cross terms on the
fp8 matrix pipe (gfx950: v_mfma_scale_f32_32x32x64_f8f6f4, operands e4m3 with one power-of-two scale per 32 elements along k).  CPU emulation
in numpy: fp16 (hi, lo) pairs as the kernels form them, e4m3 by round-to-nearest on a 3-bit mantissa with the block's shared exponent,
fp32 accumulation.  Prints the error of the three-pass product and of the fp8-cross-term product against float64, relative to the largest
|result| (the convention of the 1e-3 tolerance), for activation / weight statistics like the model's."""
import numpy as np


def split_f16(x):
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float16).astype(np.float32)
    return hi, lo


def mx_e4m3(v, axis):
    """e4m3 with a shared power-of-two scale per 32 elements along `axis` (OCP MX): scale = 2^(floor(log2 max|v|) - 8), e4m3 max 448"""
    v = np.moveaxis(v, axis, -1)
    sh = v.shape
    b = v.reshape(sh[:-1] + (sh[-1] // 32, 32))
    amax = np.abs(b).max(-1, keepdims=True)
    e = np.floor(np.log2(np.where(amax > 0, amax, 1.0))) - 8.0
    s = np.exp2(e)
    y = b / s
    # e4m3: 3 mantissa bits, normal exponents -6 .. 8, subnormals below 2^-6
    ay = np.abs(y)
    ex = np.clip(np.floor(np.log2(np.where(ay > 0, ay, 1.0))), -6, 8)
    q = np.exp2(ex - 3)
    r = np.round(y / q) * q
    r = np.clip(r, -448.0, 448.0)
    return np.moveaxis((r * s).reshape(sh), -1, axis).astype(np.float32)


def run(M, K, N, xs, ws, seed):
    g = np.random.default_rng(seed)
    x = (g.standard_normal((M, K)) * xs).astype(np.float32)
    x[:, : K // 8] *= 30.0                                   # a few large features (log-mel-like rows after the embedding)
    W = (g.standard_normal((K, N)) * ws).astype(np.float32)
    ref = x.astype(np.float64) @ W.astype(np.float64)
    xh, xl = split_f16(x); Wh, Wl = split_f16(W)
    three = (xh @ Wh).astype(np.float32) + (xh @ Wl) + (xl @ Wh)
    cross8 = (mx_e4m3(xh, 1) @ mx_e4m3(Wl, 0)) + (mx_e4m3(xl, 1) @ mx_e4m3(Wh, 0))
    mixed = (xh @ Wh).astype(np.float32) + cross8
    den = np.abs(ref).max()
    return np.abs(three - ref).max() / den, np.abs(mixed - ref).max() / den, np.abs((xh @ Wh) - ref).max() / den


if __name__ == '__main__':
    print('%-34s %12s %12s %12s' % ('shape / scales', 'three passes', 'fp8 cross', 'hi.hi only'))
    for (M, K, N, xs, ws) in ((512, 256, 512, 1.0, 0.06), (512, 512, 256, 0.5, 0.04), (512, 256, 768, 3.0, 0.06), (512, 768, 256, 1e-4, 0.06)):
        e3, e8, e1 = np.mean([run(M, K, N, xs, ws, s) for s in range(3)], axis=0)
        print('%-34s %12.2e %12.2e %12.2e' % ('%d x %d x %d, x %.0e, W %.0e' % (M, K, N, xs, ws), e3, e8, e1))
