// HBM-bound helper kernels of the hFT-Transformer path (gfx950): weight preparation, encoder-front fold,
// window materialisation, LayerNorm backward, decoder transposes, output heads, fused loss, fused Adam.
// All of them stream fp32 with 16-byte accesses where the layout allows; reductions are two-stage and
// deterministic (no float atomics).  Entry points and reference citations: include/hftt_hip.h.
#include "hftt_common.h"
#include "x3_common.h"
#include "hftt_host.h"
#include "../../include/hftt_hip.h"
#include <math.h>

namespace {

// one fp32 value -> the two 16-bit halves of the split ("x3") operand form
template <int E>
__device__ __forceinline__ void x3_split1(float x, uint16_t& hi, uint16_t& lo) {
  unsigned h, l;
  x3_split2_checked<E>(x, 0.f, h, l);      // (weight preparation: a NaN / Inf parameter must poison the planes)
  hi = (uint16_t)(h & 0xFFFFu); lo = (uint16_t)(l & 0xFFFFu);
}

// ------------------------------------------------------------------ weight preparation
__global__ void prep_weights_kernel(const float* __restrict__ params, uint16_t* __restrict__ wbf, float* __restrict__ wf32,
                                    float* __restrict__ fdst, const hftt_prep_entry* __restrict__ table) {
  const hftt_prep_entry e = table[blockIdx.x];
  const long total = (long)e.rows * e.cols;
  for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.y * blockDim.x) {
    const int r = (int)(i / e.cols), c = (int)(i % e.cols);
    const float x = params[e.src_off + (long)r * e.src_ld + c];
    if (e.kind == 2) {
      fdst[e.dst_off + (long)r * e.dst_ld + c] = x;
    } else {
      const long d = (e.kind == 1) ? (e.dst_off + (long)c * e.dst_ld + r) : (e.dst_off + (long)r * e.dst_ld + c);
      if (wbf != nullptr) wbf[d] = f2bf(x);
      if (wf32 != nullptr) wf32[d] = x;
    }
  }
}

// the same table, matrices as two 16-bit planes (hi + lo ~ the fp32 value); e.pad = element type (2 fp16, 4 bf16)
__global__ void prep_weights_x3_kernel(const float* __restrict__ params, uint16_t* __restrict__ whi, uint16_t* __restrict__ wlo,
                                       float* __restrict__ fdst, const hftt_prep_entry* __restrict__ table) {
  const hftt_prep_entry e = table[blockIdx.x];
  const long total = (long)e.rows * e.cols;
  for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.y * blockDim.x) {
    const int r = (int)(i / e.cols), c = (int)(i % e.cols);
    const float x = params[e.src_off + (long)r * e.src_ld + c];
    if (e.kind == 2) {
      fdst[e.dst_off + (long)r * e.dst_ld + c] = x;
    } else {
      const long d = (e.kind == 1) ? (e.dst_off + (long)c * e.dst_ld + r) : (e.dst_off + (long)r * e.dst_ld + c);
      uint16_t hi, lo;
      if (e.pad == X3_BF16) x3_split1<X3_BF16>(x, hi, lo); else x3_split1<X3_F16>(x, hi, lo);
      whi[d] = hi; wlo[d] = lo;
    }
  }
}

// ------------------------------------------------------------------ encoder front fold (conv + flatten + linear)
__global__ void fold_fwd_kernel(const hftt_fold_desc f) {
  const int nw = f.n_proc - f.kw + 1;
  const int cd = f.C * nw;
  const long total = (long)f.d_pad * f.Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i / f.Kp), u = (int)(i % f.Kp);
    float acc = 0.f;
    if (j < f.d && u < f.n_proc) {
      for (int c = 0; c < f.C; c++)
        for (int kk = 0; kk < f.kw; kk++) {
          const int w = u - kk;
          if (w >= 0 && w < nw) acc += f.wtok[(long)j * cd + c * nw + w] * f.wconv[c * f.kw + kk];
        }
    }
    if (f.weff_bf != nullptr) f.weff_bf[i] = f2bf(acc);
    if (f.weff_f32 != nullptr) f.weff_f32[i] = acc;
    if (f.weff_hi != nullptr) x3_split1<X3_F16>(acc, f.weff_hi[i], f.weff_lo[i]);
    if (u == 0 && j < f.d) {
      float b = f.btok[j];
      for (int c = 0; c < f.C; c++) {
        float s = 0.f;
        for (int w = 0; w < nw; w++) s += f.wtok[(long)j * cd + c * nw + w];
        b += f.bconv[c] * s;
      }
      f.beff[j] = b;
    }
  }
}

// grads of tok_embedding_freq.{weight,bias}
__global__ void fold_bwd_tok_kernel(const hftt_fold_desc f) {
  const int nw = f.n_proc - f.kw + 1;
  const int cd = f.C * nw;
  const long total = (long)f.d * cd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i / cd), cw = (int)(i % cd);
    const int c = cw / nw, w = cw % nw;
    float acc = f.dbeff[j] * f.bconv[c];
    for (int kk = 0; kk < f.kw; kk++) acc += f.dweff[(long)j * f.Kp + w + kk] * f.wconv[c * f.kw + kk];
    f.g_wtok[i] = acc;
    if (cw == 0) f.g_btok[j] = f.dbeff[j];
  }
}
// grads of conv.{weight,bias}: one workgroup per (c,kk) plus one per c for the bias
__global__ __launch_bounds__(256) void fold_bwd_conv_kernel(const hftt_fold_desc f) {
  __shared__ float red[256];
  const int nw = f.n_proc - f.kw + 1;
  const int cd = f.C * nw;
  const int b = blockIdx.x;
  const bool is_bias = b >= f.C * f.kw;
  const int c = is_bias ? (b - f.C * f.kw) : (b / f.kw);
  const int kk = is_bias ? 0 : (b % f.kw);
  float acc = 0.f;
  for (int i = threadIdx.x; i < f.d * nw; i += 256) {
    const int j = i / nw, w = i % nw;
    const float wt = f.wtok[(long)j * cd + c * nw + w];
    acc += is_bias ? f.dbeff[j] * wt : f.dweff[(long)j * f.Kp + w + kk] * wt;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (is_bias) f.g_bconv[c] = red[0];
    else f.g_wconv[c * f.kw + kk] = red[0];
  }
}

// A[(b,t,f), u] = spec[b, f, t+u]
__global__ void im2win_kernel(const float* __restrict__ spec, float* __restrict__ win, int B, int F, int T, int n_proc, int Kp) {
  const int W = T + n_proc - 1;
  const int q4 = Kp / 4;
  const long total = (long)B * T * F * q4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int u4 = (int)(i % q4);
    const long row = i / q4;
    const int f = (int)(row % F);
    const int t = (int)((row / F) % T);
    const int b = (int)(row / ((long)F * T));
    const float* src = spec + ((long)b * F + f) * W + t;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int u = u4 * 4 + e;
      v[e] = (u < n_proc) ? src[u] : 0.f;
    }
    *reinterpret_cast<float4*>(win + row * Kp + u4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ------------------------------------------------------------------ LayerNorm backward
template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const hftt_ln_bwd_desc g) {
  constexpr int N = VPL * 64;
  __shared__ float red[4][2][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const bool dy_bf = g.io_flags & HFTT_LNB_DY_BF16, dr_bf = g.io_flags & HFTT_LNB_DR_BF16, r_bf = g.io_flags & HFTT_LNB_R_BF16;
  float gam[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int e = 0; e < VPL; e++) { gam[e] = g.gamma[lane * VPL + e]; dg[e] = 0.f; db[e] = 0.f; }
  for (long row = (long)blockIdx.x * 4 + wave; row < g.M; row += (long)gridDim.x * 4) {
    float dy[VPL], r[VPL];
    const long base = row * N + lane * VPL;
    if (VPL == 4) {
      const float4 a = hftt_load4(g.dy, dy_bf, base);
      const float4 b = hftt_load4(g.r, r_bf, base);
      dy[0] = a.x; dy[1 % VPL] = a.y; dy[2 % VPL] = a.z; dy[3 % VPL] = a.w;
      r[0] = b.x; r[1 % VPL] = b.y; r[2 % VPL] = b.z; r[3 % VPL] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < VPL; e++) { dy[e] = dy_bf ? bf2f(reinterpret_cast<const unsigned short*>(g.dy)[base + e]) : g.dy[base + e]; r[e] = r_bf ? bf2f(reinterpret_cast<const unsigned short*>(g.r)[base + e]) : g.r[base + e]; }
    }
    const float mean = g.mean[row], rstd = g.rstd[row];
    float xh[VPL], gg[VPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < VPL; e++) {
      xh[e] = (r[e] - mean) * rstd;
      gg[e] = dy[e] * gam[e];
      s1 += gg[e];
      s2 += gg[e] * xh[e];
      dg[e] += dy[e] * xh[e];
      db[e] += dy[e];
    }
    s1 = wave_sum(s1) * (1.0f / N);
    s2 = wave_sum(s2) * (1.0f / N);
    float o[VPL], od[VPL];
#pragma unroll
    for (int e = 0; e < VPL; e++) {
      o[e] = rstd * (gg[e] - s1 - xh[e] * s2);
      od[e] = o[e];
      if (g.dr_drop != nullptr && g.drop_p > 0.f)
        od[e] = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)(base + e), thr) ? o[e] * inv_keep : 0.f;
    }
    if (VPL == 4) {
      hftt_store4(g.dr, dr_bf, base, o[0], o[1 % VPL], o[2 % VPL], o[3 % VPL]);
      if (g.dr_drop != nullptr) {
        if (g.drop_bf16) {
          uint2 u;
          u.x = f2bf(od[0]) | ((unsigned)f2bf(od[1 % VPL]) << 16); u.y = f2bf(od[2 % VPL]) | ((unsigned)f2bf(od[3 % VPL]) << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.dr_drop) + base) = u;
        } else {
          *reinterpret_cast<float4*>(g.dr_drop + base) = make_float4(od[0], od[1 % VPL], od[2 % VPL], od[3 % VPL]);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < VPL; e++) {
        hftt_store1(g.dr, dr_bf, base + e, o[e]);
        if (g.dr_drop != nullptr) {
          if (g.drop_bf16) reinterpret_cast<unsigned short*>(g.dr_drop)[base + e] = f2bf(od[e]);
          else g.dr_drop[base + e] = od[e];
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VPL; e++) { red[wave][0][lane * VPL + e] = dg[e]; red[wave][1][lane * VPL + e] = db[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += 256) {
    const int which = i / N, c = i % N;
    g.ws[(long)blockIdx.x * 2 * N + i] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// N = 64 (the reference's default width): a wave takes FOUR rows per step -- 16 lanes x 4 elements per row, one 16-byte (fp32) or 8-byte
// (bf16) access per lane and tensor, a 16-lane reduction -- instead of one row of 4-byte accesses and a 64-lane reduction: the row-per-wave
// form above sat at 3.4 TB/s at this width (13 launches, 0.66 ms of the tiny configuration's 5.4 ms step).
__global__ __launch_bounds__(256) void ln_bwd64_kernel(const hftt_ln_bwd_desc g) {
  constexpr int N = 64;
  __shared__ float red[4][2][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4, c0 = (lane & 15) * 4;
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const bool dy_bf = g.io_flags & HFTT_LNB_DY_BF16, dr_bf = g.io_flags & HFTT_LNB_DR_BF16, r_bf = g.io_flags & HFTT_LNB_R_BF16;
  const float4 gm = *reinterpret_cast<const float4*>(g.gamma + c0);
  const float gam[4] = {gm.x, gm.y, gm.z, gm.w};
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  for (long row0 = ((long)blockIdx.x * 4 + wave) * 4; row0 < g.M; row0 += (long)gridDim.x * 16) {
    const long row = row0 + sub;
    const bool ok = row < g.M;                       // (uniform over the 16 lanes of a row)
    const long rc = ok ? row : (long)g.M - 1;
    const long base = rc * N + c0;
    const float4 a = hftt_load4(g.dy, dy_bf, base);
    const float4 b = hftt_load4(g.r, r_bf, base);
    const float mean = g.mean[rc], rstd = g.rstd[rc];
    const float dy[4] = {ok ? a.x : 0.f, ok ? a.y : 0.f, ok ? a.z : 0.f, ok ? a.w : 0.f}, r[4] = {b.x, b.y, b.z, b.w};
    float xh[4], gg[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      xh[e] = (r[e] - mean) * rstd;
      gg[e] = dy[e] * gam[e];
      s1 += gg[e];
      s2 += gg[e] * xh[e];
      dg[e] += dy[e] * xh[e];
      db[e] += dy[e];
    }
    s1 = group_sum<16>(s1) * (1.0f / N);
    s2 = group_sum<16>(s2) * (1.0f / N);
    float o[4], od[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      o[e] = rstd * (gg[e] - s1 - xh[e] * s2);
      od[e] = o[e];
      if (g.dr_drop != nullptr && g.drop_p > 0.f)
        od[e] = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)(base + e), thr) ? o[e] * inv_keep : 0.f;
    }
    if (ok) {
      hftt_store4(g.dr, dr_bf, base, o[0], o[1], o[2], o[3]);
      if (g.dr_drop != nullptr) {
        if (g.drop_bf16) {
          uint2 u;
          u.x = f2bf(od[0]) | ((unsigned)f2bf(od[1]) << 16); u.y = f2bf(od[2]) | ((unsigned)f2bf(od[3]) << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.dr_drop) + base) = u;
        } else {
          *reinterpret_cast<float4*>(g.dr_drop + base) = make_float4(od[0], od[1], od[2], od[3]);
        }
      }
    }
  }
  // the four row groups of the wave hold partial sums of the same columns
#pragma unroll
  for (int e = 0; e < 4; e++) {
    dg[e] += __shfl_xor(dg[e], 16); dg[e] += __shfl_xor(dg[e], 32);
    db[e] += __shfl_xor(db[e], 16); db[e] += __shfl_xor(db[e], 32);
  }
  if (lane < 16) {
#pragma unroll
    for (int e = 0; e < 4; e++) { red[wave][0][c0 + e] = dg[e]; red[wave][1][c0 + e] = db[e]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += 256) {
    const int which = i / N, c = i % N;
    g.ws[(long)blockIdx.x * 2 * N + i] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// N = 256 with dy, r, dr (and the dropped copy) all stored as bf16 -- the bf16 gradient stream of the strip plans.  The row-per-wave form
// above moves 8 bytes per lane and per tensor and goes through a 64-lane reduction for every row, one row at a time: it sat at 3.7 TB/s.
// Here a wave takes FOUR rows per step (16 lanes x 16 elements per row: two 16-byte loads per lane and tensor, a 16-lane reduction) and the
// next step's loads are issued before the current step is computed.
__device__ __forceinline__ void unpack16(const uint4& a, const uint4& b, float* v) {
  const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 8; e++) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u); }
}
__device__ __forceinline__ void pack16(const float* v, uint4& a, uint4& b) {
  unsigned w[8];
#pragma unroll
  for (int e = 0; e < 8; e++) w[e] = f2bf(v[2 * e]) | ((unsigned)f2bf(v[2 * e + 1]) << 16);
  a = make_uint4(w[0], w[1], w[2], w[3]); b = make_uint4(w[4], w[5], w[6], w[7]);
}
__global__ __launch_bounds__(256) void ln_bwd256_bf16_kernel(const hftt_ln_bwd_desc g) {
  constexpr int N = 256;
  __shared__ float red[16][2][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4, cl = lane & 15;               // row within the group of four, 16-column chunk
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const bool drop = g.dr_drop != nullptr && g.drop_p > 0.f;
  const unsigned short* dyp = reinterpret_cast<const unsigned short*>(g.dy);
  const unsigned short* rp = reinterpret_cast<const unsigned short*>(g.r);
  unsigned short* drp = reinterpret_cast<unsigned short*>(g.dr);
  unsigned short* ddp = reinterpret_cast<unsigned short*>(g.dr_drop);
  float gam[16], dg[16], db[16];
#pragma unroll
  for (int e = 0; e < 16; e++) { gam[e] = g.gamma[cl * 16 + e]; dg[e] = 0.f; db[e] = 0.f; }
  const long ngrp = ((long)g.M + 3) / 4;
  const long stride = (long)gridDim.x * 4;
  long grp = (long)blockIdx.x * 4 + wave;
  uint4 y0, y1, r0, r1;
  float mean, rstd;
  auto load = [&](long gq) {
    const long row = gq * 4 + sub;
    const long rc = row < g.M ? row : (long)g.M - 1;       // clamped: loads stay unconditional
    const uint4* py = reinterpret_cast<const uint4*>(dyp + rc * N + cl * 16);
    const uint4* pr = reinterpret_cast<const uint4*>(rp + rc * N + cl * 16);
    y0 = py[0]; y1 = py[1]; r0 = pr[0]; r1 = pr[1];
    mean = g.mean[rc]; rstd = g.rstd[rc];
  };
  if (grp < ngrp) load(grp);
  for (; grp < ngrp; grp += stride) {
    float dy[16], xh[16];
    unpack16(y0, y1, dy);
    unpack16(r0, r1, xh);
    const float mu = mean, rs = rstd;
    const long row = grp * 4 + sub;
    const long nxt = grp + stride;
    load(nxt < ngrp ? nxt : ngrp - 1);                     // the next step's rows are in flight while this step is computed
    const bool valid = row < g.M;
    float gg[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      if (!valid) dy[e] = 0.f;
      xh[e] = (xh[e] - mu) * rs;
      gg[e] = dy[e] * gam[e];
      s1 += gg[e];
      s2 += gg[e] * xh[e];
      dg[e] += dy[e] * xh[e];
      db[e] += dy[e];
    }
    s1 = group_sum<16>(s1) * (1.0f / N);
    s2 = group_sum<16>(s2) * (1.0f / N);
    float o[16];
#pragma unroll
    for (int e = 0; e < 16; e++) o[e] = rs * (gg[e] - s1 - xh[e] * s2);
    if (valid) {
      uint4 a, b;
      pack16(o, a, b);
      uint4* po = reinterpret_cast<uint4*>(drp + row * N + cl * 16);
      po[0] = a; po[1] = b;
      if (g.dr_drop != nullptr) {
        if (drop) {
          const uint64_t q0 = (uint64_t)(row * N + cl * 16) >> 2;              // four hash quads (hftt_keep)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t k4 = hftt_keep_quad(g.drop_seed, g.drop_site, q0 + q, thr);
#pragma unroll
            for (int f = 0; f < 4; f++) o[4 * q + f] = ((k4 >> f) & 1u) ? o[4 * q + f] * inv_keep : 0.f;
          }
          pack16(o, a, b);
        }
        uint4* pd = reinterpret_cast<uint4*>(ddp + row * N + cl * 16);
        pd[0] = a; pd[1] = b;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; e++) { red[wave * 4 + sub][0][cl * 16 + e] = dg[e]; red[wave * 4 + sub][1][cl * 16 + e] = db[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += 256) {
    const int which = i / N, c = i % N;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) acc += red[q][which][c];                    // fixed order: reproducible
    g.ws[(long)blockIdx.x * 2 * N + i] = acc;
  }
}

// N = 256, any mix of fp32 / bf16 storage (the x3 plans: fp32 dy and dr, bf16 saved sums): R rows per wave, the lane's columns taken as
// groups of four at a stride of 4 * (64 / R) so that the lanes of a row cover CONTIGUOUS bytes with every access; PF: the next step's rows
// are in flight while the current step is computed.  Measured at 262,144 rows in that storage mix (tools/bench_ln_bwd.py): the row-per-wave
// kernel above (64-lane reduction with four v_readlane per sum) 5.1 TB/s, R = 4 4.2 .. 4.4 TB/s (16 elements per lane in four arrays:
// registers cost it the occupancy that hides the latency), R = 2 5.3 TB/s with or without PF -- the form instantiated.
template <int R, bool PF>
__global__ __launch_bounds__(256) void ln_bwd256_rows_kernel(const hftt_ln_bwd_desc g) {
  constexpr int N = 256, L = 64 / R, G = N / (4 * L), E = 4 * G;      // lanes per row, float4 groups and elements per lane
  __shared__ float red[4 * R][2][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / L, cl = lane % L;
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const bool drop = g.dr_drop != nullptr && g.drop_p > 0.f;
  const bool dy_bf = g.io_flags & HFTT_LNB_DY_BF16, dr_bf = g.io_flags & HFTT_LNB_DR_BF16, r_bf = g.io_flags & HFTT_LNB_R_BF16;
  float gam[E], dg[E], db[E];
#pragma unroll
  for (int k = 0; k < G; k++) {
    const float4 t = *reinterpret_cast<const float4*>(g.gamma + cl * 4 + 4 * L * k);
    gam[4 * k] = t.x; gam[4 * k + 1] = t.y; gam[4 * k + 2] = t.z; gam[4 * k + 3] = t.w;
  }
#pragma unroll
  for (int e = 0; e < E; e++) { dg[e] = 0.f; db[e] = 0.f; }
  const long ngrp = ((long)g.M + R - 1) / R;
  const long stride = (long)gridDim.x * 4;
  long grp = (long)blockIdx.x * 4 + wave;
  float4 yv[G], rv[G];
  float mean, rstd;
  auto load = [&](long gq) {
    const long row = gq * R + sub;
    const long rc = row < g.M ? row : (long)g.M - 1;       // clamped: loads stay unconditional
    const long base = rc * N + cl * 4;
#pragma unroll
    for (int k = 0; k < G; k++) { yv[k] = hftt_load4(g.dy, dy_bf, base + 4 * L * k); rv[k] = hftt_load4(g.r, r_bf, base + 4 * L * k); }
    mean = g.mean[rc]; rstd = g.rstd[rc];
  };
  if (PF && grp < ngrp) load(grp);
  for (; grp < ngrp; grp += stride) {
    if (!PF) load(grp);
    float dy[E], xh[E];
#pragma unroll
    for (int k = 0; k < G; k++) {
      dy[4 * k] = yv[k].x; dy[4 * k + 1] = yv[k].y; dy[4 * k + 2] = yv[k].z; dy[4 * k + 3] = yv[k].w;
      xh[4 * k] = rv[k].x; xh[4 * k + 1] = rv[k].y; xh[4 * k + 2] = rv[k].z; xh[4 * k + 3] = rv[k].w;
    }
    const float mu = mean, rs = rstd;
    const long row = grp * R + sub;
    if (PF) {
      const long nxt = grp + stride;
      load(nxt < ngrp ? nxt : ngrp - 1);                   // the next step's rows are in flight while this step is computed
    }
    const bool valid = row < g.M;
    float gg[E], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (!valid) dy[e] = 0.f;
      xh[e] = (xh[e] - mu) * rs;
      gg[e] = dy[e] * gam[e];
      s1 += gg[e];
      s2 += gg[e] * xh[e];
      dg[e] += dy[e] * xh[e];
      db[e] += dy[e];
    }
    s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    if (L == 32) { s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16); }
    s1 *= (1.0f / N); s2 *= (1.0f / N);
    float o[E];
#pragma unroll
    for (int e = 0; e < E; e++) o[e] = rs * (gg[e] - s1 - xh[e] * s2);
    if (valid) {
      const long base = row * N + cl * 4;
#pragma unroll
      for (int k = 0; k < G; k++) hftt_store4(g.dr, dr_bf, base + 4 * L * k, o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
      if (g.dr_drop != nullptr) {
#pragma unroll
        for (int k = 0; k < G; k++) {
          float d4[4] = {o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]};
          if (drop) {
            const uint32_t k4 = hftt_keep_quad(g.drop_seed, g.drop_site, (uint64_t)(base + 4 * L * k) >> 2, thr);      // one hash quad (hftt_keep)
#pragma unroll
            for (int f = 0; f < 4; f++) d4[f] = ((k4 >> f) & 1u) ? d4[f] * inv_keep : 0.f;
          }
          hftt_store4(g.dr_drop, g.drop_bf16 != 0, base + 4 * L * k, d4[0], d4[1], d4[2], d4[3]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < G; k++)
#pragma unroll
    for (int f = 0; f < 4; f++) {
      red[wave * R + sub][0][cl * 4 + 4 * L * k + f] = dg[4 * k + f];
      red[wave * R + sub][1][cl * 4 + 4 * L * k + f] = db[4 * k + f];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += 256) {
    const int which = i / N, c = i % N;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 4 * R; q++) acc += red[q][which][c];                    // fixed order: reproducible
    g.ws[(long)blockIdx.x * 2 * N + i] = acc;
  }
}

// out[c] = beta*out[c] + sum_w ws[w][c]   (c over 2N entries: dgamma then dbeta)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ ws, int n_wg, int N, float* dgamma, float* dbeta, float beta) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float acc = 0.f;
  if (c < 2 * N) {
    // eight independent loads in flight per thread (the plain loop was one dependent L2 round trip per partial: ~20 us for 2 MB)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = rg;
    for (; w + 16 * 7 < n_wg; w += 16 * 8) {
#pragma unroll
      for (int u = 0; u < 8; u++) a8[u] += ws[(long)(w + 16 * u) * 2 * N + c];
    }
    for (; w < n_wg; w += 16) a8[0] += ws[(long)w * 2 * N + c];
    acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  }
  red[rg][cl] = acc;
  __syncthreads();
  if (rg == 0 && c < 2 * N) {
    float s = 0.f;
    for (int q = 0; q < 16; q++) s += red[q][cl];
    float* dst = (c < N) ? (dgamma + c) : (dbeta + (c - N));
    *dst = (beta != 0.f) ? (*dst * beta + s) : s;
  }
}

// ------------------------------------------------------------------ decoder time transpose (+scale, +pos, dropout)
__global__ void time_embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos, float* __restrict__ y,
                                      int B, int T, int Nn, int d, float scale, float drop_p, uint32_t site, uint64_t seed, uint32_t io_flags) {
  const bool x_bf = io_flags & HFTT_TE_X_BF16, y_bf = io_flags & HFTT_TE_Y_BF16;
  const int d4 = d / 4;
  const long total = (long)B * Nn * T * d4;
  const uint32_t thr = hftt_keep_thr(drop_p);
  const float inv_keep = hftt_keep_scale(drop_p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4);
    const long orow = i / d4;                      // (b*Nn + n)*T + t
    const int t = (int)(orow % T);
    const int n = (int)((orow / T) % Nn);
    const int b = (int)(orow / ((long)T * Nn));
    const long irow = ((long)b * T + t) * Nn + n;
    const float4 a = hftt_load4(x, x_bf, irow * d + c4 * 4);
    const float4 p = *reinterpret_cast<const float4*>(pos + (long)t * d + c4 * 4);
    float v[4] = {a.x * scale + p.x, a.y * scale + p.y, a.z * scale + p.z, a.w * scale + p.w};
    if (drop_p > 0.f) {
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = hftt_keep(seed, site, (uint64_t)(orow * d + c4 * 4 + e), thr) ? v[e] * inv_keep : 0.f;
    }
    hftt_store4(y, y_bf, orow * d + c4 * 4, v[0], v[1], v[2], v[3]);
  }
}
__global__ void time_embed_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dym,
                                      int B, int T, int Nn, int d, float scale, float drop_p, uint32_t site, uint64_t seed, int accumulate, uint32_t io_flags) {
  const bool dy_bf = io_flags & HFTT_TE_X_BF16, dx_bf = io_flags & HFTT_TE_Y_BF16, dym_bf = io_flags & HFTT_TE_M_BF16;
  const int d4 = d / 4;
  const long total = (long)B * Nn * T * d4;
  const uint32_t thr = hftt_keep_thr(drop_p);
  const float inv_keep = hftt_keep_scale(drop_p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4);
    const long orow = i / d4;
    const int t = (int)(orow % T);
    const int n = (int)((orow / T) % Nn);
    const int b = (int)(orow / ((long)T * Nn));
    const long irow = ((long)b * T + t) * Nn + n;
    const float4 a = hftt_load4(dy, dy_bf, orow * d + c4 * 4);
    float v[4] = {a.x, a.y, a.z, a.w};
    if (drop_p > 0.f) {
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = hftt_keep(seed, site, (uint64_t)(orow * d + c4 * 4 + e), thr) ? v[e] * inv_keep : 0.f;
    }
    if (dym != nullptr) hftt_store4(dym, dym_bf, orow * d + c4 * 4, v[0], v[1], v[2], v[3]);
    float4 o = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
    if (accumulate) { const float4 old = hftt_load4(dx, dx_bf, irow * d + c4 * 4); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
    hftt_store4(dx, dx_bf, irow * d + c4 * 4, o.x, o.y, o.z, o.w);
  }
}
__global__ void dropout_bwd_kernel(float* __restrict__ gbuf, long n, float drop_p, uint32_t site, uint64_t seed, uint32_t bf) {
  const uint32_t thr = hftt_keep_thr(drop_p);
  const float inv_keep = hftt_keep_scale(drop_p);
  const long n4 = n >> 2;                          // n % 4 == 0 (host check): 4 consecutive elements = one hash quad per thread
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = hftt_load4(gbuf, bf, i * 4);
    float v[4] = {a.x, a.y, a.z, a.w};
    const uint32_t k4 = hftt_keep_quad(seed, site, (uint64_t)i, thr);
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = ((k4 >> e) & 1u) ? v[e] * inv_keep : 0.f;
    hftt_store4(gbuf, bf, i * 4, v[0], v[1], v[2], v[3]);
  }
}

// ------------------------------------------------------------------ column sums (two-stage)
constexpr int CS_SPLITS = 16;
__global__ void colsum_stage1_kernel(const float* __restrict__ x, long rows, long n, long ld, float* __restrict__ ws, uint32_t bf) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int sp = blockIdx.y;
  const long per = (rows + CS_SPLITS - 1) / CS_SPLITS;
  const long r0 = sp * per;
  long r1 = r0 + per; if (r1 > rows) r1 = rows;
  float acc = 0.f;
  if (bf) { for (long r = r0; r < r1; r++) acc += bf2f(reinterpret_cast<const unsigned short*>(x)[r * ld + j]); }
  else { for (long r = r0; r < r1; r++) acc += x[r * ld + j]; }
  ws[(long)sp * n + j] = acc;
}
__global__ void colsum_stage2_kernel(const float* __restrict__ ws, long n, float* __restrict__ out, float beta) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float acc = 0.f;
#pragma unroll
  for (int s = 0; s < CS_SPLITS; s++) acc += ws[(long)s * n + j];
  out[j] = (beta != 0.f) ? (out[j] * beta + acc) : acc;
}

// ------------------------------------------------------------------ output heads
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void heads_split_kernel(const float* __restrict__ logits, long ldl, float* __restrict__ onset, float* __restrict__ offset,
                                   float* __restrict__ mpe, float* __restrict__ velocity, int B, int T, int Nn, int V, int time_major) {
  const int v4 = V / 4;
  const long total = (long)B * T * Nn * v4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % v4);
    const long s = i / v4;
    long o = s;
    if (time_major) {
      const int t = (int)(s % T);
      const int n = (int)((s / T) % Nn);
      const int b = (int)(s / ((long)T * Nn));
      o = ((long)b * T + t) * Nn + n;
    }
    *reinterpret_cast<float4*>(velocity + o * V + c4 * 4) = *reinterpret_cast<const float4*>(logits + s * ldl + c4 * 4);
    if (c4 == 0) {
      onset[o] = sigmoidf_(logits[s * ldl + V]);
      offset[o] = sigmoidf_(logits[s * ldl + V + 1]);
      mpe[o] = sigmoidf_(logits[s * ldl + V + 2]);
    }
  }
}
__global__ void heads_split_bwd_kernel(const float* __restrict__ p_on, const float* __restrict__ p_of, const float* __restrict__ p_mp,
                                       const float* __restrict__ d_on, const float* __restrict__ d_of, const float* __restrict__ d_mp,
                                       const float* __restrict__ d_vel, float* __restrict__ dlogits, long ldl,
                                       int B, int T, int Nn, int V, int time_major) {
  const int l4 = (int)(ldl / 4);
  const int v4 = V / 4;
  const long total = (long)B * T * Nn * l4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % l4);
    const long s = i / l4;
    long o = s;
    if (time_major) {
      const int t = (int)(s % T);
      const int n = (int)((s / T) % Nn);
      const int b = (int)(s / ((long)T * Nn));
      o = ((long)b * T + t) * Nn + n;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < v4) {
      v = *reinterpret_cast<const float4*>(d_vel + o * V + c4 * 4);
    } else if (c4 == v4) {
      const float a = p_on[o], b2 = p_of[o], c = p_mp[o];
      v.x = d_on[o] * a * (1.f - a);
      v.y = d_of[o] * b2 * (1.f - b2);
      v.z = d_mp[o] * c * (1.f - c);
    }
    *reinterpret_cast<float4*>(dlogits + s * ldl + c4 * 4) = v;
  }
}

// ------------------------------------------------------------------ fused loss (6x BCE + 2x CE) and its gradient
constexpr int LOSS_WGS = 4096;
__global__ __launch_bounds__(256) void loss_kernel(const hftt_loss_desc g) {
  __shared__ float red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_n = 1.0f / (float)g.n;
  float part = 0.f;    // lanes 0..5: BCE terms; lanes 6,7: CE terms (A, B)
  for (long el = (long)blockIdx.x * 4 + wave; el < g.n; el += (long)gridDim.x * 4) {
    if (lane < 6) {
      const float p = g.prob[lane][el];
      const float y = (lane % 3 == 0) ? g.label_onset[el] : ((lane % 3 == 1) ? g.label_offset[el] : g.label_mpe[el]);
      const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(log1pf(-p), -100.f);
      part += -(y * lp + (1.f - y) * l1p);
      if (g.d_prob[lane] != nullptr) {
        const float w = (lane < 3) ? g.weight_A : g.weight_B;
        g.d_prob[lane][el] = (p - y) / fmaxf((1.f - p) * p, 1e-12f) * (w * inv_n * g.grad_scale);
      }
    }
    const int label = (int)g.label_velocity[el];
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const float* lg = g.vel[side] + el * g.V;
      float v[4], mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int c = lane + 64 * e;
        v[e] = (c < g.V) ? lg[c] : -INFINITY;
        mx = fmaxf(mx, v[e]);
      }
      mx = wave_max(mx);
      float s = 0.f, ex[4];
#pragma unroll
      for (int e = 0; e < 4; e++) { ex[e] = (lane + 64 * e < g.V) ? expf(v[e] - mx) : 0.f; s += ex[e]; }
      s = wave_sum(s);
      const float lse = mx + logf(s);
      float picked = 0.f;
#pragma unroll
      for (int e = 0; e < 4; e++) if (lane + 64 * e == label) picked = v[e];
      picked = wave_sum(picked);
      if (lane == 6 + side) part += lse - picked;
      if (g.d_vel[side] != nullptr) {
        const float w = (side == 0 ? g.weight_A : g.weight_B) * inv_n * g.grad_scale;
        float* dst = g.d_vel[side] + el * g.V;
        const float inv_s = 1.0f / s;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int c = lane + 64 * e;
          if (c < g.V) dst[c] = (ex[e] * inv_s - (c == label ? 1.f : 0.f)) * w;
        }
      }
    }
  }
  if (lane < 8) red[wave][lane] = part;
  __syncthreads();
  if (threadIdx.x < 8) g.ws[(long)blockIdx.x * 8 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// V <= 128, V % 4 == 0 (the model's 128 velocity classes): half a wave per element, four classes per lane -- 16-byte loads and stores, two
// elements per wave step (the general kernel above moves 4 bytes per lane and keeps 58 lanes idle in its BCE part: 1.5 TB/s).
__device__ __forceinline__ float half_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
  v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64));
  return v;
}
__global__ __launch_bounds__(256) void loss_v4_kernel(const hftt_loss_desc g) {
  __shared__ float red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = lane & 31, hf = lane >> 5;
  const float inv_n = 1.0f / (float)g.n;
  float part = 0.f;    // per half: lanes 0..5 BCE terms; lanes 6,7 CE terms (A, B)
  for (long el0 = ((long)blockIdx.x * 4 + wave) * 2; el0 < g.n; el0 += (long)gridDim.x * 8) {
    const long el = el0 + hf;
    const bool ok = el < g.n;                       // (odd n: the second half of the last pair idles, but takes part in the shuffles)
    const long elc = ok ? el : g.n - 1;
    if (l32 < 6 && ok) {
      const float p = g.prob[l32][el];
      const float y = (l32 % 3 == 0) ? g.label_onset[el] : ((l32 % 3 == 1) ? g.label_offset[el] : g.label_mpe[el]);
      const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(log1pf(-p), -100.f);
      part += -(y * lp + (1.f - y) * l1p);
      if (g.d_prob[l32] != nullptr) {
        const float w = (l32 < 3) ? g.weight_A : g.weight_B;
        g.d_prob[l32][el] = (p - y) / fmaxf((1.f - p) * p, 1e-12f) * (w * inv_n * g.grad_scale);
      }
    }
    const int label = (int)g.label_velocity[elc];
    const int c0 = 4 * l32;
    const bool cok = c0 < g.V;
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const float* lg = g.vel[side] + elc * g.V;
      float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (cok) v = *reinterpret_cast<const float4*>(lg + c0);
      const float mx = half_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
      float4 ex = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cok) ex = make_float4(expf(v.x - mx), expf(v.y - mx), expf(v.z - mx), expf(v.w - mx));
      const float s = half_sum((ex.x + ex.y) + (ex.z + ex.w));
      const float lse = mx + logf(s);
      const int k = label - c0;
      float picked = (k == 0) ? v.x : (k == 1) ? v.y : (k == 2) ? v.z : (k == 3) ? v.w : 0.f;
      picked = half_sum(picked);
      if (l32 == 6 + side && ok) part += lse - picked;
      if (g.d_vel[side] != nullptr && cok && ok) {
        const float w = (side == 0 ? g.weight_A : g.weight_B) * inv_n * g.grad_scale;
        const float inv_s = 1.0f / s;
        float4 o;
        o.x = (ex.x * inv_s - (k == 0 ? 1.f : 0.f)) * w; o.y = (ex.y * inv_s - (k == 1 ? 1.f : 0.f)) * w;
        o.z = (ex.z * inv_s - (k == 2 ? 1.f : 0.f)) * w; o.w = (ex.w * inv_s - (k == 3 ? 1.f : 0.f)) * w;
        *reinterpret_cast<float4*>(g.d_vel[side] + el * g.V + c0) = o;
      }
    }
  }
  part += __shfl_xor(part, 32, 64);                 // the two halves' terms
  if (lane < 8) red[wave][lane] = part;
  __syncthreads();
  if (threadIdx.x < 8) g.ws[(long)blockIdx.x * 8 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void loss_reduce_kernel(const hftt_loss_desc g, int n_wg) {
  __shared__ float red[32][8];
  const int term = threadIdx.x & 7, grp = threadIdx.x >> 3;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four independent chains (fixed order: the sum is reproducible), loads in flight together
  int w = grp;
  for (; w + 96 < n_wg; w += 128) {
    const float v0 = g.ws[(long)w * 8 + term], v1 = g.ws[(long)(w + 32) * 8 + term], v2 = g.ws[(long)(w + 64) * 8 + term], v3 = g.ws[(long)(w + 96) * 8 + term];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; w < n_wg; w += 32) a0 += g.ws[(long)w * 8 + term];
  red[grp][term] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 8) {
    float s = 0.f;
    for (int q = 0; q < 32; q++) s += red[q][threadIdx.x];
    red[0][threadIdx.x] = s / (float)g.n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // order of loss_out[1..8]: onset_A, offset_A, mpe_A, velocity_A, onset_B, offset_B, mpe_B, velocity_B
    const float t[8] = {red[0][0], red[0][1], red[0][2], red[0][6], red[0][3], red[0][4], red[0][5], red[0][7]};
    float la = (t[0] + t[1]) + t[2];
    la += t[3];
    float lb = (t[4] + t[5]) + t[6];
    lb += t[7];
    g.loss_out[0] = g.weight_A * la + g.weight_B * lb;
    for (int i = 0; i < 8; i++) g.loss_out[1 + i] = t[i];
  }
}

// ------------------------------------------------------------------ fused Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                            float lr_c, float beta1, float beta2, float omb1, float omb2, float eps, float inv_sqrt_bc2, float grad_scale) {
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pe = &pp.x; const float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float gr = ge[e] * grad_scale;
      me[e] = beta1 * me[e] + omb1 * gr;
      ve[e] = beta2 * ve[e] + omb2 * gr * gr;
      pe[e] -= lr_c * me[e] / (sqrtf(ve[e]) * inv_sqrt_bc2 + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  const long tail = n4 * 4;
  for (long i = tail + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gr = g[i] * grad_scale;
    m[i] = beta1 * m[i] + omb1 * gr;
    v[i] = beta2 * v[i] + omb2 * gr * gr;
    p[i] -= lr_c * m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps);
  }
}

inline int grid_for(long work_items, int block, int cap = 4096) {
  long b = (work_items + block - 1) / block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int hftt_prep_weights(const float* params, uint16_t* wbf, float* wf32, float* fdst,
                                 const hftt_prep_entry* table_dev, int n_entries, void* stream) {
  HFTT_REQUIRE(params && (wbf || wf32 || fdst) && table_dev && n_entries > 0, "prep_weights: null operand");
  hipLaunchKernelGGL(prep_weights_kernel, dim3((unsigned)n_entries, 48), dim3(256), 0, (hipStream_t)stream, params, wbf, wf32, fdst, table_dev);
  HFTT_CHECK_LAUNCH("prep_weights");
  return 0;
}

extern "C" int hftt_prep_weights_x3(const float* params, uint16_t* whi, uint16_t* wlo, float* fdst,
                                    const hftt_prep_entry* table_dev, int n_entries, void* stream) {
  HFTT_REQUIRE(params && whi && wlo && table_dev && n_entries > 0, "prep_weights_x3: null operand");
  hipLaunchKernelGGL(prep_weights_x3_kernel, dim3((unsigned)n_entries, 48), dim3(256), 0, (hipStream_t)stream, params, whi, wlo, fdst, table_dev);
  HFTT_CHECK_LAUNCH("prep_weights_x3");
  return 0;
}

extern "C" int hftt_embed_fold_fwd(const hftt_fold_desc* d, void* stream) {
  HFTT_REQUIRE(d && d->wconv && d->bconv && d->wtok && d->btok && (d->weff_bf || d->weff_f32 || d->weff_hi) && d->beff, "embed_fold_fwd: null operand");
  HFTT_REQUIRE((d->weff_hi == nullptr) == (d->weff_lo == nullptr), "embed_fold_fwd: the split planes come as a pair");
  HFTT_REQUIRE(d->Kp % 32 == 0 && d->Kp >= d->n_proc && d->d_pad >= d->d && d->n_proc >= d->kw, "embed_fold_fwd: bad shape");
  hipLaunchKernelGGL(fold_fwd_kernel, dim3(grid_for((long)d->d_pad * d->Kp, 256)), dim3(256), 0, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("embed_fold_fwd");
  return 0;
}
extern "C" int hftt_embed_fold_bwd(const hftt_fold_desc* d, void* stream) {
  HFTT_REQUIRE(d && d->dweff && d->dbeff && d->g_wconv && d->g_bconv && d->g_wtok && d->g_btok, "embed_fold_bwd: null operand");
  const int nw = d->n_proc - d->kw + 1;
  hipLaunchKernelGGL(fold_bwd_tok_kernel, dim3(grid_for((long)d->d * d->C * nw, 256)), dim3(256), 0, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("embed_fold_bwd(tok)");
  hipLaunchKernelGGL(fold_bwd_conv_kernel, dim3((unsigned)(d->C * d->kw + d->C)), dim3(256), 0, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("embed_fold_bwd(conv)");
  return 0;
}
extern "C" int hftt_im2win(const float* spec, float* win, int32_t B, int32_t F, int32_t T, int32_t n_proc, int32_t Kp, void* stream) {
  HFTT_REQUIRE(spec && win && B > 0 && F > 0 && T > 0 && n_proc > 0 && Kp % 4 == 0 && Kp >= n_proc, "im2win: bad arguments");
  hipLaunchKernelGGL(im2win_kernel, dim3(grid_for((long)B * T * F * (Kp / 4), 256, 8192)), dim3(256), 0, (hipStream_t)stream, spec, win, B, F, T, n_proc, Kp);
  HFTT_CHECK_LAUNCH("im2win");
  return 0;
}

extern "C" int32_t hftt_ln_bwd_wgs(int32_t M) {
  int w = (M + 3) / 4;
  if (w > 1024) w = 1024;
  if (w < 1) w = 1;
  return w;
}
extern "C" int hftt_ln_bwd(const hftt_ln_bwd_desc* d, void* stream) {
  HFTT_REQUIRE(d && d->dy && d->r && d->mean && d->rstd && d->gamma && d->dr && d->ws, "ln_bwd: null operand");
  HFTT_REQUIRE(d->N == 64 || d->N == 128 || d->N == 256, "ln_bwd: N=%d must be 64, 128 or 256", d->N);
  const int wgs = hftt_ln_bwd_wgs(d->M);
  const uint32_t all_bf = HFTT_LNB_DY_BF16 | HFTT_LNB_DR_BF16 | HFTT_LNB_R_BF16;
  const bool fast = d->N == 256 && (d->io_flags & all_bf) == all_bf && (d->dr_drop == nullptr || d->drop_bf16) &&
                    ((((uintptr_t)d->dy | (uintptr_t)d->r | (uintptr_t)d->dr | (uintptr_t)d->dr_drop) & 15) == 0);
  if (fast) hipLaunchKernelGGL(ln_bwd256_bf16_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, *d);
  else if (d->N == 256 && ((((uintptr_t)d->dy | (uintptr_t)d->r | (uintptr_t)d->dr | (uintptr_t)d->dr_drop | (uintptr_t)d->gamma) & 15) == 0))
    hipLaunchKernelGGL((ln_bwd256_rows_kernel<2, true>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, *d);
  else if (d->N == 256) hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, *d);
  else if (d->N == 128) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, *d);
  else if (((((uintptr_t)d->dy | (uintptr_t)d->r | (uintptr_t)d->dr | (uintptr_t)d->dr_drop | (uintptr_t)d->gamma) & 15) == 0))
    hipLaunchKernelGGL(ln_bwd64_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("ln_bwd");
  return 0;
}
extern "C" int hftt_ln_bwd_reduce(const float* ws, int32_t n_wg, int32_t N, float* dgamma, float* dbeta, float beta, void* stream) {
  HFTT_REQUIRE(ws && dgamma && dbeta && n_wg > 0 && N > 0, "ln_bwd_reduce: bad arguments");
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * N + 15) / 16), dim3(256), 0, (hipStream_t)stream, ws, n_wg, N, dgamma, dbeta, beta);
  HFTT_CHECK_LAUNCH("ln_bwd_reduce");
  return 0;
}

extern "C" int hftt_time_embed_fwd(const float* x, const float* pos, float* y, int32_t B, int32_t T, int32_t Nn, int32_t d,
                                   float scale, float drop_p, uint32_t site, uint64_t seed, uint32_t io_flags, void* stream) {
  HFTT_REQUIRE(x && pos && y && d % 4 == 0, "time_embed_fwd: bad arguments");
  hipLaunchKernelGGL(time_embed_fwd_kernel, dim3(grid_for((long)B * T * Nn * (d / 4), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     x, pos, y, B, T, Nn, d, scale, drop_p, site, seed, io_flags);
  HFTT_CHECK_LAUNCH("time_embed_fwd");
  return 0;
}
extern "C" int hftt_time_embed_bwd(const float* dy, float* dx, float* dym, int32_t B, int32_t T, int32_t Nn, int32_t d,
                                   float scale, float drop_p, uint32_t site, uint64_t seed, int32_t accumulate, uint32_t io_flags, void* stream) {
  HFTT_REQUIRE(dy && dx && d % 4 == 0, "time_embed_bwd: bad arguments");
  hipLaunchKernelGGL(time_embed_bwd_kernel, dim3(grid_for((long)B * T * Nn * (d / 4), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     dy, dx, dym, B, T, Nn, d, scale, drop_p, site, seed, accumulate, io_flags);
  HFTT_CHECK_LAUNCH("time_embed_bwd");
  return 0;
}
extern "C" int hftt_dropout_bwd(float* g, int64_t n, float drop_p, uint32_t site, uint64_t seed, uint32_t bf16, void* stream) {
  HFTT_REQUIRE(g && n > 0 && n % 4 == 0 && drop_p >= 0.f && drop_p < 1.f && ((uintptr_t)g & 15) == 0, "dropout_bwd: bad arguments (n %% 4 == 0, 16-byte aligned)");
  if (drop_p == 0.f) return 0;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, g, (long)n, drop_p, site, seed, bf16);
  HFTT_CHECK_LAUNCH("dropout_bwd");
  return 0;
}
extern "C" int64_t hftt_colsum_ws_bytes(int64_t rows, int64_t n) { (void)rows; return (int64_t)CS_SPLITS * n * 4; }
extern "C" int hftt_colsum(const float* x, int64_t rows, int64_t n, int64_t ld, float* out, float beta, float* ws, uint32_t x_bf16, void* stream) {
  HFTT_REQUIRE(x && out && ws && rows > 0 && n > 0 && ld >= n, "colsum: bad arguments");
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3((unsigned)((n + 255) / 256), CS_SPLITS), dim3(256), 0, (hipStream_t)stream, x, (long)rows, (long)n, (long)ld, ws, x_bf16);
  HFTT_CHECK_LAUNCH("colsum(1)");
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, (long)n, out, beta);
  HFTT_CHECK_LAUNCH("colsum(2)");
  return 0;
}
extern "C" int hftt_heads_split(const float* logits, int64_t ldl, float* onset, float* offset, float* mpe, float* velocity,
                                int32_t B, int32_t T, int32_t Nn, int32_t V, int32_t time_major, void* stream) {
  HFTT_REQUIRE(logits && onset && offset && mpe && velocity, "heads_split: null operand");
  HFTT_REQUIRE(V % 4 == 0 && ldl % 4 == 0 && ldl >= V + 3, "heads_split: V=%d / ldl=%ld unsupported", V, (long)ldl);
  hipLaunchKernelGGL(heads_split_kernel, dim3(grid_for((long)B * T * Nn * (V / 4), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     logits, (long)ldl, onset, offset, mpe, velocity, B, T, Nn, V, time_major);
  HFTT_CHECK_LAUNCH("heads_split");
  return 0;
}
extern "C" int hftt_heads_split_bwd(const float* p_onset, const float* p_offset, const float* p_mpe,
                                    const float* d_onset, const float* d_offset, const float* d_mpe, const float* d_velocity,
                                    float* dlogits, int64_t ldl, int32_t B, int32_t T, int32_t Nn, int32_t V, int32_t time_major, void* stream) {
  HFTT_REQUIRE(p_onset && p_offset && p_mpe && d_onset && d_offset && d_mpe && d_velocity && dlogits, "heads_split_bwd: null operand");
  HFTT_REQUIRE(V % 4 == 0 && ldl % 4 == 0 && ldl >= V + 4, "heads_split_bwd: V=%d / ldl=%ld unsupported", V, (long)ldl);
  hipLaunchKernelGGL(heads_split_bwd_kernel, dim3(grid_for((long)B * T * Nn * (ldl / 4), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     p_onset, p_offset, p_mpe, d_onset, d_offset, d_mpe, d_velocity, dlogits, (long)ldl, B, T, Nn, V, time_major);
  HFTT_CHECK_LAUNCH("heads_split_bwd");
  return 0;
}

extern "C" int64_t hftt_loss_ws_bytes(int64_t n) { (void)n; return (int64_t)LOSS_WGS * 8 * 4; }
extern "C" int hftt_loss(const hftt_loss_desc* d, void* stream) {
  HFTT_REQUIRE(d && d->n > 0 && d->V > 0 && d->V <= 256, "loss: bad shape");
  for (int i = 0; i < 6; i++) HFTT_REQUIRE(d->prob[i] != nullptr, "loss: null probability tensor %d", i);
  HFTT_REQUIRE(d->vel[0] && d->vel[1] && d->label_onset && d->label_offset && d->label_mpe && d->label_velocity && d->loss_out && d->ws, "loss: null operand");
  int wgs = (int)((d->n + 3) / 4);
  const bool v4 = d->V <= 128 && d->V % 4 == 0 && (((uintptr_t)d->vel[0] | (uintptr_t)d->vel[1] | (uintptr_t)d->d_vel[0] | (uintptr_t)d->d_vel[1]) & 15) == 0;
  if (v4) wgs = (int)((d->n + 7) / 8);
  if (wgs > LOSS_WGS) wgs = LOSS_WGS;
  if (v4) hipLaunchKernelGGL(loss_v4_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL(loss_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("loss");
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *d, wgs);
  HFTT_CHECK_LAUNCH("loss_reduce");
  return 0;
}

extern "C" int hftt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int32_t step,
                              double lr, double beta1, double beta2, double eps, double grad_scale, void* stream) {
  HFTT_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  HFTT_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
  // hyper-parameters arrive as doubles (python floats) and 1-beta is formed BEFORE rounding to fp32, as torch.optim.Adam does: 1.f - 0.999f is
  // off by 1.3e-5 relative and would show in exp_avg_sq of every step
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float lr_c = (float)(lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                     lr_c, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, inv_sqrt_bc2, (float)grad_scale);
  HFTT_CHECK_LAUNCH("adam_step");
  return 0;
}
