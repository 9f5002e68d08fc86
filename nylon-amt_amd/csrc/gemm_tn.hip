// TN GEMM (weight gradients) on MFMA 32x32x16 bf16 (gfx950):
//   dW[N,K] = out_scale * dY[M,N]^T . X[M,K],   db[N] = colsum(dY)      -- include/hftt_hip.h (hftt_gemm_tn)
// The reduction runs over the (huge) token dimension M, so both operands are needed "token-major" in the
// MFMA fragments.  Tiles of dY and X are staged ROW-major in LDS (coalesced 16-byte global loads) and the
// fragments are fetched with ds_read_b64_tr_b16 (hardware transpose; lane map verified by tools/probe_mfma T5).
// npass == 3 (parity): tiles stay fp32 in LDS and the product runs on v_mfma_f32_32x32x2_f32 (one f32 per lane per
// operand, so the row-major tile IS the fragment layout: plain conflict-free ds_read_b32, no transpose needed).
// M is split over workgroups; partial tiles go to a slab workspace and a second kernel reduces them in a fixed
// order (bitwise reproducible; no float atomics).
#include <stdlib.h>
#include "hftt_common.h"
#include "hftt_host.h"
#include "../../include/hftt_hip.h"

namespace {

constexpr int BMT = 32;   // token rows per LDS stage

struct TnPlan {
  int tm, tn;            // template tile selectors
  int tile_n, tile_k;    // output tile
  int n_tiles, k_tiles, splits, rows_per_split;
  long nws, kws;         // padded slab dims
};

TnPlan tn_plan(int M, int N, int K) {
  TnPlan p;
  static int force_small = -1;
  if (force_small < 0) { const char* e = getenv("HFTT_TN_SMALL"); force_small = (e && e[0] == '1') ? 1 : 0; }
  if (N >= 256 && K >= 256 && !force_small) { p.tm = 2; p.tn = 4; }
  else if (N >= 128 && K >= 128) { p.tm = 1; p.tn = 2; }
  else { p.tm = 1; p.tn = 1; }
  p.tile_n = 128 * p.tm;
  p.tile_k = 64 * p.tn;
  p.n_tiles = (N + p.tile_n - 1) / p.tile_n;
  p.k_tiles = (K + p.tile_k - 1) / p.tile_k;
  const int tiles = p.n_tiles * p.k_tiles;
  int target = (p.tm == 2) ? 256 : 512;
  int splits = target / tiles;
  if (splits < 1) splits = 1;
  const int max_splits = (M + 63) / 64;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rps = (M + splits - 1) / splits;
  rps = ((rps + BMT - 1) / BMT) * BMT;
  p.splits = (M + rps - 1) / rps;
  p.rows_per_split = rps;
  p.nws = (long)p.n_tiles * p.tile_n;
  p.kws = (long)p.k_tiles * p.tile_k;
  return p;
}

template <int TM, int TN, int NPASS>
struct TnCfg {
  static constexpr bool F32 = (NPASS == 3);
  static constexpr int TILE_N = 128 * TM;
  static constexpr int TILE_K = 64 * TN;
  // bf16: row stride in shorts, bytes = 2*TILE + 64 == 64 (mod 128): conflict-free tr reads.  fp32: floats, 16-byte aligned rows.
  static constexpr int RSY = F32 ? TILE_N + 4 : TILE_N + 32;
  static constexpr int RSX = F32 ? TILE_K + 4 : TILE_K + 32;
  static constexpr int ESZ = F32 ? 4 : 2;
  static constexpr int Y_ELEMS = BMT * RSY;
  static constexpr int X_ELEMS = BMT * RSX;
  static constexpr int BUF_ELEMS = Y_ELEMS + X_ELEMS;
  static constexpr int LDS_BYTES = 2 * BUF_ELEMS * ESZ;
  static constexpr int YL = TILE_N / 64;    // float4 loads per thread (dY)
  static constexpr int XL = (TILE_K + 63) / 64;   // float4 loads per thread (X)
};

template <int TM, int TN, int NPASS>
__global__ __launch_bounds__(512) void gemm_tn_kernel(const hftt_gemm_tn_desc g, const int k_tiles, const int rows_per_split,
                                                     const long nws, const long kws) {
  using Cfg = TnCfg<TM, TN, NPASS>;
  constexpr bool F32 = Cfg::F32;
  constexpr int RSY = Cfg::RSY, RSX = Cfg::RSX, TILE_N = Cfg::TILE_N, TILE_K = Cfg::TILE_K;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* sm16 = reinterpret_cast<unsigned short*>(smem);
  float* sm32 = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn4 = wave >> 1, wk2 = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x;
  const int nt = tile / k_tiles, ktile = tile % k_tiles;
  const int n0 = nt * TILE_N, k0 = ktile * TILE_K;
  const int split = blockIdx.y;
  const long mbeg = (long)split * rows_per_split;
  long mend = mbeg + rows_per_split;
  if (mend > g.M) mend = g.M;
  const int nsteps = (int)((mend - mbeg + BMT - 1) / BMT);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  float4 yregA[Cfg::YL], xregA[Cfg::XL], yregB[Cfg::YL], xregB[Cfg::XL];   // two staging sets: loads run two steps ahead
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool dy_bf = !F32 && (g.io_flags & HFTT_TN_DY_BF16), x_bf = !F32 && (g.io_flags & HFTT_TN_X_BF16);
  constexpr int YF4R = TILE_N / 4;   // float4 per row
  constexpr int XF4R = TILE_K / 4;
  const int yc4 = tid % YF4R;
  const int ycol = n0 + yc4 * 4;
  const bool ycol_ok = ycol < g.N;    // N % 4 == 0
  const int ycol_c = ycol_ok ? ycol : g.N - 4;

  auto gload = [&](int step, float4 (&yreg)[Cfg::YL], float4 (&xreg)[Cfg::XL]) {
    const long mb = mbeg + (long)step * BMT;
#pragma unroll
    for (int j = 0; j < Cfg::YL; j++) {
      const int i = tid + 512 * j;
      const int row = i / YF4R;
      const long m = mb + row;
      // unconditional load from a clamped address + select: a branch around the load would make hipcc wait vmcnt(0) per load
      const long mc = m < mend ? m : mend - 1;
      float4 t;
      if (dy_bf) {      // 4 bf16 = 8 bytes (exact widening; the tile is re-rounded to the same bf16 values on the way into LDS)
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(g.dY) + mc * g.lddy + ycol_c);
        t = make_float4(bf2f(u.x & 0xFFFFu), bf2f(u.x >> 16), bf2f(u.y & 0xFFFFu), bf2f(u.y >> 16));
      } else {
        t = *reinterpret_cast<const float4*>(g.dY + mc * g.lddy + ycol_c);
      }
      yreg[j] = (ycol_ok && m < mend) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < Cfg::XL; j++) {
      const int i = tid + 512 * j;
      const int row = i / XF4R, c4 = i % XF4R;
      const long m = mb + row;
      const int col = k0 + c4 * 4;
      const long mc = m < mend ? m : mend - 1;
      const int cc = col < g.K ? col : g.K - 4;
      float4 t;
      if (x_bf) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(g.X) + mc * g.ldx + cc);
        t = make_float4(bf2f(u.x & 0xFFFFu), bf2f(u.x >> 16), bf2f(u.y & 0xFFFFu), bf2f(u.y >> 16));
      } else {
        t = *reinterpret_cast<const float4*>(g.X + mc * g.ldx + cc);
      }
      xreg[j] = (row < BMT && col < g.K && m < mend) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto put4 = [&](int buf, int which, int off, const float4& f) {      // which: 0 = dY tile, 1 = X tile
    if (F32) {
      float* base = sm32 + buf * Cfg::BUF_ELEMS + which * Cfg::Y_ELEMS;
      *reinterpret_cast<float4*>(base + off) = f;
    } else {
      unsigned short* base = sm16 + buf * Cfg::BUF_ELEMS + which * Cfg::Y_ELEMS;
      uint2 ph;
      ph.x = f2bf(f.x) | ((unsigned)f2bf(f.y) << 16); ph.y = f2bf(f.z) | ((unsigned)f2bf(f.w) << 16);
      *reinterpret_cast<uint2*>(base + off) = ph;
    }
  };
  auto sstore = [&](int buf, const float4 (&yreg)[Cfg::YL], const float4 (&xreg)[Cfg::XL]) {
#pragma unroll
    for (int j = 0; j < Cfg::YL; j++) {
      const int i = tid + 512 * j;
      const int row = i / YF4R;
      csum[0] += yreg[j].x; csum[1] += yreg[j].y; csum[2] += yreg[j].z; csum[3] += yreg[j].w;
      put4(buf, 0, row * RSY + yc4 * 4, yreg[j]);
    }
#pragma unroll
    for (int j = 0; j < Cfg::XL; j++) {
      const int i = tid + 512 * j;
      const int row = i / XF4R, c4 = i % XF4R;
      if (row < BMT) put4(buf, 1, row * RSX + c4 * 4, xreg[j]);
    }
  };

  // transposed-fragment addressing for the bf16 path (probe T5): 16-lane group gi, block row qq, column quad p
  const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int frag_col = 16 * (gi & 1) + 4 * pp;   // column inside a 32-wide tile
  const int frag_row = 8 * (gi >> 1) + qq;       // + 16*s + 4*half

  if (nsteps > 0) {
    gload(0, yregA, xregA);
    sstore(0, yregA, xregA);
    if (nsteps > 1) gload(1, yregB, xregB);
  }
  __syncthreads();
  // step s computes from LDS buffer (s&1); set `cur` (stored to LDS one step ago) is refilled with step s+2, set `nxt`
  // (holding step s+1, issued one step ago) is converted into the other LDS buffer after the MFMAs
  auto body = [&](int step, float4 (&ycur)[Cfg::YL], float4 (&xcur)[Cfg::XL], const float4 (&ynxt)[Cfg::YL], const float4 (&xnxt)[Cfg::XL]) {
    const int buf = step & 1;
    if (step + 2 < nsteps) gload(step + 2, ycur, xcur);
    if (F32) {
      const float* Ys = sm32 + buf * Cfg::BUF_ELEMS;
      const float* Xs = Ys + Cfg::Y_ELEMS;
#pragma unroll
      for (int t = 0; t < 16; t++) {
        const int m = 16 * lh + t;          // token row handled by this lane half in MFMA step t
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) a[i] = Ys[m * RSY + wn4 * TM * 32 + i * 32 + lr];
#pragma unroll
        for (int j = 0; j < TN; j++) b[j] = Xs[m * RSX + wk2 * TN * 32 + j * 32 + lr];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = mfma32_f32(a[i], b[j], acc[i][j]);
      }
    } else {
      const unsigned short* Ys = sm16 + buf * Cfg::BUF_ELEMS;
      const unsigned short* Xs = Ys + Cfg::Y_ELEMS;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) {
          const int c = wn4 * TM * 32 + i * 32 + frag_col;
          const int r0 = 16 * s + frag_row;
          a[i] = join4(lds_read_tr16(Ys + r0 * RSY + c), lds_read_tr16(Ys + (r0 + 4) * RSY + c));
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
          const int c = wk2 * TN * 32 + j * 32 + frag_col;
          const int r0 = 16 * s + frag_row;
          b[j] = join4(lds_read_tr16(Xs + r0 * RSX + c), lds_read_tr16(Xs + (r0 + 4) * RSX + c));
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
      }
    }
    if (step + 1 < nsteps) sstore(buf ^ 1, ynxt, xnxt);
    __syncthreads();
  };
  for (int step = 0; step < nsteps; step += 2) {
    body(step, yregA, xregA, yregB, xregB);
    if (step + 1 < nsteps) body(step + 1, yregB, xregB, yregA, xregA);
  }

  // partial tile -> slab [split][nws][kws]
  float* slab = reinterpret_cast<float*>(g.ws) + (long)split * nws * kws;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const long n = n0 + wn4 * TM * 32 + i * 32 + acc_row32(r, lh);
        const long k = k0 + wk2 * TN * 32 + j * 32 + lr;
        slab[n * kws + k] = acc[i][j][r];
      }
  // column sums of dY (bias gradient): only the k-tile-0 workgroups publish them
  if (ktile == 0) {
    float* red = reinterpret_cast<float*>(smem);   // [512/YF4R][TILE_N]
    __syncthreads();
    const int grp = tid / YF4R;
#pragma unroll
    for (int e = 0; e < 4; e++) red[grp * TILE_N + yc4 * 4 + e] = csum[e];
    __syncthreads();
    float* bslab = reinterpret_cast<float*>(g.ws) + (long)gridDim.y * nws * kws + (long)split * nws;
    for (int c = tid; c < TILE_N; c += 512) {
      float s = 0.f;
      for (int q = 0; q < 512 / YF4R; q++) s += red[q * TILE_N + c];
      bslab[n0 + c] = s;
    }
  }
}

// Slab reduce: one thread per output element, splits summed in a FIXED order (bitwise reproducible) with 8 loads in flight.
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const hftt_gemm_tn_desc g, const int splits, const long nws, const long kws,
                                                             const int nb_main) {
  const long total = (long)g.N * g.K_out;
  const float* ws = reinterpret_cast<const float*>(g.ws);
  const long sstride = nws * kws;
  if ((int)blockIdx.x >= nb_main) {
    // bias rows: one wave per output row n, lanes stride over the splits, fixed-order tree reduce
    const int n = ((int)blockIdx.x - nb_main) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= g.N) return;
    int sg = -1;
    for (int s = 0; s < g.n_seg; s++)
      if (n >= g.seg_row0[s] && n < g.seg_row0[s] + g.seg_rows[s]) sg = s;
    if (sg < 0 || g.seg_db[sg] == nullptr) return;
    const float* bs = ws + (long)splits * sstride;
    float b = 0.f;
    for (int q = lane; q < splits; q += 64) b += bs[(long)q * nws + n];
    b = wave_sum(b);
    if (lane == 0) {
      float* db = g.seg_db[sg] + (n - g.seg_row0[sg]);
      const float bv = b * g.out_scale;
      *db = (g.beta != 0.f) ? (*db * g.beta + bv) : bv;
    }
    return;
  }
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)nb_main * blockDim.x) {
    const int n = (int)(idx / g.K_out), k = (int)(idx % g.K_out);
    int sg = -1;
    for (int s = 0; s < g.n_seg; s++)
      if (n >= g.seg_row0[s] && n < g.seg_row0[s] + g.seg_rows[s]) sg = s;
    if (sg < 0) continue;
    const float* p = ws + (long)n * kws + k;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int s = 0;
    for (; s + 8 <= splits; s += 8) {
      const float v0 = p[(long)(s + 0) * sstride], v1 = p[(long)(s + 1) * sstride], v2 = p[(long)(s + 2) * sstride], v3 = p[(long)(s + 3) * sstride];
      const float v4 = p[(long)(s + 4) * sstride], v5 = p[(long)(s + 5) * sstride], v6 = p[(long)(s + 6) * sstride], v7 = p[(long)(s + 7) * sstride];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6; a7 += v7;
    }
    for (; s < splits; s++) a0 += p[(long)s * sstride];
    const float acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    float* dst = g.seg_dw[sg] + (long)(n - g.seg_row0[sg]) * g.K_out + k;
    const float v = acc * g.out_scale;
    *dst = (g.beta != 0.f) ? (*dst * g.beta + v) : v;
  }
}

template <int TM, int TN, int NPASS>
int launch_tn(const hftt_gemm_tn_desc& d, const TnPlan& p, hipStream_t st) {
  using Cfg = TnCfg<TM, TN, NPASS>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<TM, TN, NPASS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) { hftt_set_error("gemm_tn: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  dim3 grid((unsigned)(p.n_tiles * p.k_tiles), (unsigned)p.splits, 1);
  hipLaunchKernelGGL((gemm_tn_kernel<TM, TN, NPASS>), grid, dim3(512), Cfg::LDS_BYTES, st, d, p.k_tiles, p.rows_per_split, p.nws, p.kws);
  HFTT_CHECK_LAUNCH("gemm_tn");
  return 0;
}

}  // namespace

extern "C" int64_t hftt_gemm_tn_ws_bytes(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const TnPlan p = tn_plan(M, N, K);
  return (int64_t)p.splits * p.nws * (p.kws + 1) * 4;
}

extern "C" int hftt_gemm_tn(const hftt_gemm_tn_desc* d, void* stream) {
  HFTT_REQUIRE(d != nullptr, "gemm_tn: null descriptor");
  HFTT_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_tn: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  HFTT_REQUIRE(d->N % 4 == 0 && d->K % 4 == 0, "gemm_tn: N=%d and K=%d must be multiples of 4", d->N, d->K);
  HFTT_REQUIRE(d->lddy % 4 == 0 && d->ldx % 4 == 0, "gemm_tn: leading dims must be multiples of 4");
  HFTT_REQUIRE(((uintptr_t)d->dY & 15) == 0 && ((uintptr_t)d->X & 15) == 0, "gemm_tn: dY/X must be 16-byte aligned");
  HFTT_REQUIRE(d->npass == 1 || d->npass == 3, "gemm_tn: npass must be 1 or 3");
  HFTT_REQUIRE(d->io_flags == 0 || d->npass == 1, "gemm_tn: bf16-stored operands need npass == 1");
  HFTT_REQUIRE(d->n_seg >= 1 && d->n_seg <= 4, "gemm_tn: n_seg must be 1..4");
  HFTT_REQUIRE(d->K_out > 0 && d->K_out <= d->K, "gemm_tn: K_out out of range");
  for (int s = 0; s < d->n_seg; s++) {
    HFTT_REQUIRE(d->seg_dw[s] != nullptr, "gemm_tn: null segment pointer");
    HFTT_REQUIRE(d->seg_row0[s] >= 0 && d->seg_rows[s] > 0 && d->seg_row0[s] + d->seg_rows[s] <= d->N, "gemm_tn: segment %d out of range", s);
  }
  HFTT_REQUIRE(d->ws != nullptr && d->ws_bytes >= hftt_gemm_tn_ws_bytes(d->M, d->N, d->K), "gemm_tn: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const TnPlan p = tn_plan(d->M, d->N, d->K);
  int rc;
  if (d->npass == 3) {
    if (p.tm == 2) rc = launch_tn<2, 4, 3>(*d, p, st);
    else if (p.tn == 2) rc = launch_tn<1, 2, 3>(*d, p, st);
    else rc = launch_tn<1, 1, 3>(*d, p, st);
  } else {
    if (p.tm == 2) rc = launch_tn<2, 4, 1>(*d, p, st);
    else if (p.tn == 2) rc = launch_tn<1, 2, 1>(*d, p, st);
    else rc = launch_tn<1, 1, 1>(*d, p, st);
  }
  if (rc != 0) return rc;
  const long total = (long)d->N * d->K_out;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  const int bias_blocks = (d->N + 3) / 4;
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks + bias_blocks), dim3(256), 0, st, *d, p.splits, p.nws, p.kws, blocks);
  HFTT_CHECK_LAUNCH("gemm_tn_reduce");
  return 0;
}
