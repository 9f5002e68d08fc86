// TN GEMM (weight gradients) on MFMA 32x32x16 bf16 (gfx950):
//   dW[N,K] = out_scale * dY[M,N]^T . X[M,K],   db[N] = colsum(dY)      -- include/hftt_hip.h (hftt_gemm_tn)
// The reduction runs over the (huge) token dimension M, so both operands are needed "token-major" in the
// MFMA fragments.  Tiles of dY and X are staged ROW-major in LDS (coalesced 16-byte global loads) and the
// fragments are fetched with ds_read_b64_tr_b16 (hardware transpose; lane map verified by tools/probe_mfma T5).
// npass == 3 (parity): tiles stay fp32 in LDS and the product runs on v_mfma_f32_32x32x2_f32 (one f32 per lane per
// operand, so the row-major tile IS the fragment layout: plain conflict-free ds_read_b32, no transpose needed).
// npass == 2 / 4 ("x3", x3_common.h): fp32 operands are split into two 16-bit planes (hi, lo) on their way into LDS and every fragment
// pair feeds three MFMAs (lo.hi + hi.lo + hi.hi); npass 4 (bf16 halves) is what the weight gradients use (dY is a gradient).
// M is split over workgroups; partial tiles go to a slab workspace and a second kernel reduces them in a fixed
// order (bitwise reproducible; no float atomics).
#include <stdlib.h>
#include "hftt_common.h"
#include "x3_common.h"
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
  if (N >= 256 && K >= 256 && !force_small) {
    p.tm = 2; p.tn = 4;
    // A single 256 x 256 output tile (N, K <= 256) means 256 splits, and the 256 KB partial tile each workgroup writes to the slab -- and
    // the reduce kernel reads back -- is as large as the operands at S_n.  The 128 x 256 tile makes two tiles, so half the splits for the
    // same 256 workgroups: half the slab traffic, twice the steps per workgroup; X is read by both tiles (same XCD: its L2).  Measured
    // (tools/bench_tn.py, main + reduce): 45 -> 37 us at S_n, 80 -> 73 us at S_e.  With two or more 256-wide tiles the half tile only
    // adds re-reads (N = 768 at S_e: 145 -> 182 us), so those keep the full tile.
    if (N <= 256 && K <= 256) p.tm = 1;
  }
  else if (N >= 128 && K >= 128) { p.tm = 1; p.tn = 2; }
  else { p.tm = 1; p.tn = 1; }
  p.tile_n = 128 * p.tm;
  p.tile_k = 64 * p.tn;
  p.n_tiles = (N + p.tile_n - 1) / p.tile_n;
  p.k_tiles = (K + p.tile_k - 1) / p.tile_k;
  const int tiles = p.n_tiles * p.k_tiles;
  static int small_target = -1;
  if (small_target < 0) { const char* e = getenv("HFTT_TN_SMALL_TARGET"); small_target = e ? atoi(e) : 512; }
  int target = (p.tm == 2 || p.tn == 4) ? 256 : small_target;
  int splits = target / tiles;
  // the tiles of a split share an XCD (kernel: block -> (tile, split) map), so splits come in groups of eight and one group's
  // tiles must not exceed that XCD's share of the resident workgroups -- 33 workgroups on 32 CUs would run two rounds
  if (splits >= 8) splits = (target / 8 / tiles) * 8;
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

template <int TM, int TN, int NPASS, bool DYB = false, bool XB = false>
struct TnCfg {
  static constexpr bool F32 = (NPASS == 3);
  static constexpr bool X3M = (NPASS == 2 || NPASS == 4 || NPASS == 5 || NPASS == 6);   // 5 = npass 4 with HFTT_TN_DY_HI: dY (the gradient) enters as its bf16 rounding; 6 = npass 4 with HFTT_TN_DY_DROP: dY is masked as it is loaded
  static constexpr int PLANES = X3M ? 2 : 1;       // x3: hi plane, then lo plane of each tile
  static constexpr int YE = DYB ? 8 : 4;           // elements per 16-byte global slot (bf16- or fp32-stored operand)
  static constexpr int XE = XB ? 8 : 4;
  static constexpr int TILE_N = 128 * TM;
  static constexpr int TILE_K = 64 * TN;
  // bf16: row stride in shorts, bytes = 2*TILE + 64 == 64 (mod 128): conflict-free tr reads.  fp32: floats, 16-byte aligned rows.
  static constexpr int RSY = F32 ? TILE_N + 4 : TILE_N + 32;
  static constexpr int RSX = F32 ? TILE_K + 4 : TILE_K + 32;
  static constexpr int ESZ = F32 ? 4 : 2;
  static constexpr int Y_ELEMS = PLANES * BMT * RSY;
  static constexpr int X_ELEMS = PLANES * BMT * RSX;
  static constexpr int BUF_ELEMS = Y_ELEMS + X_ELEMS;
  static constexpr int LDS_BYTES = 2 * BUF_ELEMS * ESZ;
  static constexpr int YSPR = TILE_N / YE;         // slots per tile row
  static constexpr int XSPR = TILE_K / XE;
  static constexpr int YL = (BMT * YSPR + 511) / 512;   // 16-byte loads per thread
  static constexpr int XL = (BMT * XSPR + 511) / 512;
  static constexpr bool Y_EXACT = (BMT * YSPR) % 512 == 0;   // every thread owns a valid slot in every load round
  static constexpr bool X_EXACT = (BMT * XSPR) % 512 == 0;
};

template <int TM, int TN, int NPASS, bool DYB, bool XB>
__global__ __launch_bounds__(512) void gemm_tn_kernel(const hftt_gemm_tn_desc g, const int k_tiles, const int rows_per_split,
                                                     const long nws, const long kws, const int n_tiles_total, const int n_splits) {
  using Cfg = TnCfg<TM, TN, NPASS, DYB, XB>;
  constexpr bool F32 = Cfg::F32, X3M = Cfg::X3M;
  constexpr int EX = X3M ? ((NPASS == 5 || NPASS == 6) ? 4 : NPASS) : X3_BF16;         // element type of the split
  constexpr bool DYH = (NPASS == 5);
  constexpr bool DYD = (NPASS == 6);                 // dY = the gradient of a dropout output: the mask of (drop_p, drop_site, drop_seed) applied on load
  static_assert(!DYD || !DYB, "the masked form takes an fp32 dY");
  const uint32_t dthr = hftt_keep_thr(g.drop_p);
  const float dinv = hftt_keep_scale(g.drop_p);
  const uint64_t dkey = hftt_hash_key(g.drop_seed, g.drop_site);      // (wave-uniform: scalar registers)
  static_assert(!F32 || (!DYB && !XB), "bf16-stored operands: bf16 mode, or ONE side of a split-bf16 product");
  // x3 with a bf16-stored operand (the saved FFN hidden / its gradient, kept as bf16 for this product only): that operand IS its hi
  // half, the lo half is zero and the pass that would multiply it is skipped (two MFMAs per fragment pair instead of three)
  static_assert(!(X3M && DYB && XB), "split product with both operands bf16-stored is the plain bf16 product");
  static_assert(!(NPASS == 5 && DYB), "a bf16-stored dY is its own hi half already");
  constexpr int RSY = Cfg::RSY, RSX = Cfg::RSX, TILE_N = Cfg::TILE_N, TILE_K = Cfg::TILE_K;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* sm16 = reinterpret_cast<unsigned short*>(smem);
  float* sm32 = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn4 = wave >> 1, wk2 = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  // XCD-aware block -> (tile, split) map: workgroup L runs on XCD L % 8 (each XCD has its own L2).  The tiles of one split read
  // the same token rows (dY once per k tile, X once per n tile), so they are given the same L % 8 and adjacent dispatch slots:
  // the re-reads then hit that XCD's L2 instead of going to HBM again (PMC before: 682 MB fetched for 332 MB algorithmic).
  const int tiles = n_tiles_total;                  // 1-D grid of 8 * ceil(splits / 8) * tiles workgroups (launch_tn)
  const int L = blockIdx.x;
  const int slot = L >> 3;
  const int tile = slot % tiles;
  const int split = (L & 7) + 8 * (slot / tiles);
  if (split >= n_splits) return;                    // padding blocks of the last group of eight splits (whole workgroup leaves)
  const int nt = tile / k_tiles, ktile = tile % k_tiles;
  const int n0 = nt * TILE_N, k0 = ktile * TILE_K;
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

  uint4 yregA[Cfg::YL], xregA[Cfg::XL], yregB[Cfg::YL], xregB[Cfg::XL];   // two staging sets of raw 16-byte slots: loads run two steps ahead
  constexpr int YE = Cfg::YE, XE = Cfg::XE, YSPR = Cfg::YSPR, XSPR = Cfg::XSPR;
  float csum[YE];
#pragma unroll
  for (int e = 0; e < YE; e++) csum[e] = 0.f;
  const int ycs = tid % YSPR;                       // this thread's slot column (fixed: 512 % YSPR == 0)
  const int ycol = n0 + ycs * YE;
  const bool ycol_ok = ycol < g.N;                  // N % 8 == 0 when dY is bf16-stored, % 4 otherwise
  const int ycol_c = ycol_ok ? ycol : g.N - YE;

  auto gload = [&](int step, uint4 (&yreg)[Cfg::YL], uint4 (&xreg)[Cfg::XL]) {
    const long mb = mbeg + (long)step * BMT;
    int tid_g = tid;
    asm volatile("" : "+v"(tid_g));                  // (per-slot row / column offsets formed per call: hoisted out of the step loop they spill)
#pragma unroll
    for (int j = 0; j < Cfg::YL; j++) {
      const int i = tid_g + 512 * j;
      const int row = i / YSPR;
      const long m = mb + row;
      // unconditional load from a clamped address (a branch around the load would make hipcc wait vmcnt(0) per load); slots that
      // do not exist are zeroed when the registers are consumed (sstore) -- a select right here would wait for each load in turn
      const long mc = m < mend ? m : mend - 1;
      yreg[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(g.dY) + (mc * g.lddy + ycol_c) * (DYB ? 2 : 4));
    }
#pragma unroll
    for (int j = 0; j < Cfg::XL; j++) {
      const int i = tid_g + 512 * j;
      const int row = i / XSPR, cs = i % XSPR;
      const long m = mb + row;
      const int col = k0 + cs * XE;
      const long mc = m < mend ? m : mend - 1;
      const int cc = col < g.K ? col : g.K - XE;
      xreg[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(g.X) + (mc * g.ldx + cc) * (XB ? 2 : 4));
    }
  };
  // one 16-byte slot -> LDS tile (which: 0 = dY tile, 1 = X tile) at element offset off
  auto put_f32 = [&](int buf, int which, int off, const uint4& u) {      // fp32-stored slot: 4 elements
    const float4 f = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
    if (F32) {
      float* base = sm32 + buf * Cfg::BUF_ELEMS + which * Cfg::Y_ELEMS;
      *reinterpret_cast<float4*>(base + off) = f;
    } else if (X3M) {
      unsigned short* base = sm16 + buf * Cfg::BUF_ELEMS + which * Cfg::Y_ELEMS;
      uint2 hi, lo;
#ifdef HFTT_TN_NOSPLIT
      // timing experiment (tools/ablate_tn.sh; results garbage): the loader of VERDICT r04 item 5a -- operands that arrive pre-split only have to be
      // copied -- approximated from above: two byte permutes per four elements instead of the split's sixteen vector operations
      hi.x = __builtin_amdgcn_perm(u.y, u.x, 0x07060302u); hi.y = __builtin_amdgcn_perm(u.w, u.z, 0x07060302u); lo = hi;
#else
      x3_split4<EX>(f, hi, lo);
#endif
      *reinterpret_cast<uint2*>(base + off) = hi;
      if (!(DYH && which == 0)) *reinterpret_cast<uint2*>(base + (which ? BMT * RSX : BMT * RSY) + off) = lo;
    } else {
      unsigned short* base = sm16 + buf * Cfg::BUF_ELEMS + which * Cfg::Y_ELEMS;
      uint2 ph;
      ph.x = f2bf(f.x) | ((unsigned)f2bf(f.y) << 16); ph.y = f2bf(f.z) | ((unsigned)f2bf(f.w) << 16);
      *reinterpret_cast<uint2*>(base + off) = ph;
    }
  };
  auto put_bf16 = [&](int buf, int which, int off, const uint4& u) {     // bf16-stored slot: 8 elements, no conversion
    unsigned short* base = sm16 + buf * Cfg::BUF_ELEMS + which * Cfg::Y_ELEMS;
    *reinterpret_cast<uint4*>(base + off) = u;
  };
  auto sstore = [&](int buf, int step, const uint4 (&yreg)[Cfg::YL], const uint4 (&xreg)[Cfg::XL]) {
    const long mb = mbeg + (long)step * BMT;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int j = 0; j < Cfg::YL; j++) {
      const int i = tid + 512 * j;
      const int row = i / YSPR;
      uint4 yv = (ycol_ok && mb + row < mend) ? yreg[j] : zero4;            // slots past the split / past N count as zeros
      if (DYD && g.drop_p > 0.f) {                    // elements (m, ycol .. ycol + 3) = one hash quad of the site's [M, N] tensor
        // quad index (m * N + ycol) / 4 in 32 bits with one full-rate 24-bit multiply (host: M < 2^24, M * N / 4 < 2^32: the hash's high index word is 0)
        const uint32_t w = hftt_hash_mix(dkey, __umul24((uint32_t)(mb + row), (uint32_t)g.N >> 2) + ((uint32_t)ycol >> 2), 0u);
        yv.x = ((w & 0xFFu) < dthr) ? __float_as_uint(__uint_as_float(yv.x) * dinv) : 0u;
        yv.y = (((w >> 8) & 0xFFu) < dthr) ? __float_as_uint(__uint_as_float(yv.y) * dinv) : 0u;
        yv.z = (((w >> 16) & 0xFFu) < dthr) ? __float_as_uint(__uint_as_float(yv.z) * dinv) : 0u;
        yv.w = ((w >> 24) < dthr) ? __float_as_uint(__uint_as_float(yv.w) * dinv) : 0u;
      }
      if (Cfg::Y_EXACT || row < BMT) {      // (a per-lane branch here would also turn the register-set wait into vmcnt(0))
        if (DYB) {
          csum[0] += bf2f(yv.x & 0xFFFFu); csum[1] += bf2f(yv.x >> 16); csum[2] += bf2f(yv.y & 0xFFFFu); csum[3] += bf2f(yv.y >> 16);
          csum[4 % YE] += bf2f(yv.z & 0xFFFFu); csum[5 % YE] += bf2f(yv.z >> 16); csum[6 % YE] += bf2f(yv.w & 0xFFFFu); csum[7 % YE] += bf2f(yv.w >> 16);
          put_bf16(buf, 0, row * RSY + ycs * 8, yv);
        } else {
          csum[0] += __uint_as_float(yv.x); csum[1] += __uint_as_float(yv.y);
          csum[2] += __uint_as_float(yv.z); csum[3] += __uint_as_float(yv.w);
          put_f32(buf, 0, row * RSY + ycs * 4, yv);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < Cfg::XL; j++) {
      const int i = tid + 512 * j;
      const int row = i / XSPR, cs = i % XSPR;
      const uint4 xv = (k0 + cs * XE < g.K && mb + row < mend) ? xreg[j] : zero4;
      if (Cfg::X_EXACT || row < BMT) {
        if (XB) put_bf16(buf, 1, row * RSX + cs * 8, xv);
        else put_f32(buf, 1, row * RSX + cs * 4, xv);
      }
    }
  };

  // transposed-fragment addressing for the bf16 path (probe T5): 16-lane group gi, block row qq, column quad p
  const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int frag_col = 16 * (gi & 1) + 4 * pp;   // column inside a 32-wide tile
  const int frag_row = 8 * (gi >> 1) + qq;       // + 16*s + 4*half

  // Every load below is unconditional (rows past the split's end come back as zeros from a clamped address): a load inside a
  // conditional block makes hipcc's s_waitcnt insertion fall back to vmcnt(0), which drained the two-steps-ahead prefetch.
  gload(0, yregA, xregA);
  sstore(0, 0, yregA, xregA);
  if (!X3M) gload(1, yregB, xregB);
  __syncthreads();
  // step s computes from LDS buffer (s&1); set `cur` (stored to LDS one step ago) is refilled with step s+2, set `nxt`
  // (holding step s+1, issued one step ago) is converted into the other LDS buffer after the MFMAs
  auto body = [&](int step, uint4 (&ycur)[Cfg::YL], uint4 (&xcur)[Cfg::XL], const uint4 (&ynxt)[Cfg::YL], const uint4 (&xnxt)[Cfg::XL]) {
    const int buf = step & 1;
    // x3: ONE staging set, loads one step ahead (three times the MFMA work per step covers the latency; two sets on top of the 128
    // accumulator registers and the hi / lo fragments of the 256 x 256 tile spilled)
    gload(X3M ? step + 1 : step + 2, ycur, xcur);
    __builtin_amdgcn_sched_barrier(0);      // keep the loads up here: hipcc otherwise sinks them below the MFMAs (one step of cover, not two)
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
    } else if (X3M) {
      const unsigned short* Ys = sm16 + buf * Cfg::BUF_ELEMS;
      const unsigned short* Xs = Ys + Cfg::Y_ELEMS;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        bf16x8 ah[TM], al[TM];
#pragma unroll
        for (int i = 0; i < TM; i++) {
          const unsigned short* p = Ys + (16 * s + frag_row) * RSY + wn4 * TM * 32 + i * 32 + frag_col;
          ah[i] = join4(lds_read_tr16(p), lds_read_tr16(p + 4 * RSY));
          if (!DYB && !DYH) al[i] = join4(lds_read_tr16(p + BMT * RSY), lds_read_tr16(p + BMT * RSY + 4 * RSY));
        }
        // one X fragment pair at a time (the next pair is read under this pair's six MFMAs): all TN pairs up front, as hipcc would
        // schedule them, put the 256 x 256 tile over the register budget
        bf16x8 bh, bl;
        {
          const unsigned short* p = Xs + (16 * s + frag_row) * RSX + wk2 * TN * 32 + frag_col;
          bh = join4(lds_read_tr16(p), lds_read_tr16(p + 4 * RSX));
          if (!XB) bl = join4(lds_read_tr16(p + BMT * RSX), lds_read_tr16(p + BMT * RSX + 4 * RSX));
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
          bf16x8 nh = bh, nl = bl;
          if (j + 1 < TN) {
            const unsigned short* p = Xs + (16 * s + frag_row) * RSX + wk2 * TN * 32 + (j + 1) * 32 + frag_col;
            nh = join4(lds_read_tr16(p), lds_read_tr16(p + 4 * RSX));
            if (!XB) nl = join4(lds_read_tr16(p + BMT * RSX), lds_read_tr16(p + BMT * RSX + 4 * RSX));
          }
#pragma unroll
          for (int i = 0; i < TM; i++) {
            if (!DYB && !XB && !DYH) acc[i][j] = x3_mma<EX>(ah[i], al[i], bh, bl, acc[i][j]);
            else {                                            // same order as x3_mma: small terms first
              if (!DYB && !DYH) acc[i][j] = X3<EX>::mma(al[i], bh, acc[i][j]);
              if (!XB) acc[i][j] = X3<EX>::mma(ah[i], bl, acc[i][j]);
              acc[i][j] = X3<EX>::mma(ah[i], bh, acc[i][j]);
            }
          }
          bh = nh; bl = nl;
          __builtin_amdgcn_sched_barrier(0);
        }
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
    sstore(buf ^ 1, step + 1, ynxt, xnxt);
    __syncthreads();
  };
  for (int step = 0; step < nsteps; step += 2) {      // an odd step count runs one extra step on zeros
    if (X3M) {
      body(step, yregA, xregA, yregA, xregA);
      body(step + 1, yregA, xregA, yregA, xregA);
    } else {
      body(step, yregA, xregA, yregB, xregB);
      body(step + 1, yregB, xregB, yregA, xregA);
    }
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
    float* red = reinterpret_cast<float*>(smem);   // [512/YSPR][TILE_N]
    __syncthreads();
    const int grp = tid / YSPR;
#pragma unroll
    for (int e = 0; e < YE; e++) red[grp * TILE_N + ycs * YE + e] = csum[e];
    __syncthreads();
    float* bslab = reinterpret_cast<float*>(g.ws) + (long)n_splits * nws * kws + (long)split * nws;
    for (int c = tid; c < TILE_N; c += 512) {
      float s = 0.f;
      for (int q = 0; q < 512 / YSPR; q++) s += red[q * TILE_N + c];
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

// The same reduce for SMALL outputs (N * K_out <= 65,536: the d = 64 model's 64 x 64 .. 192 x 64 matrices).  One thread per element gives the
// launch N * K / 256 workgroups -- 16 for a 64 x 64 matrix -- to read a slab of several hundred partial tiles (15 MB): 16 us per call, 31 calls
// per tiny-model step.  Here a workgroup takes 16 consecutive elements and its 16 thread rows take the splits s = row (mod 16), four loads in
// flight; the partial sums meet in LDS and are added in a FIXED order (bitwise reproducible).  Bias rows as above.
__global__ __launch_bounds__(256) void gemm_tn_reduce_small_kernel(const hftt_gemm_tn_desc g, const int splits, const long nws, const long kws,
                                                                   const int nb_main) {
  const long total = (long)g.N * g.K_out;
  const float* ws = reinterpret_cast<const float*>(g.ws);
  const long sstride = nws * kws;
  if ((int)blockIdx.x >= nb_main) {
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
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long idx = (long)blockIdx.x * 16 + e;
  const bool ok = idx < total;
  const long idc = ok ? idx : total - 1;
  const int n = (int)(idc / g.K_out), k = (int)(idc % g.K_out);
  const float* p = ws + (long)n * kws + k;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = sl;
  for (; s + 48 < splits; s += 64) {
    const float v0 = p[(long)s * sstride], v1 = p[(long)(s + 16) * sstride], v2 = p[(long)(s + 32) * sstride], v3 = p[(long)(s + 48) * sstride];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; s < splits; s += 16) a0 += p[(long)s * sstride];
  part[sl][e] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && ok) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += part[i][e];
    int sg = -1;
    for (int q = 0; q < g.n_seg; q++)
      if (n >= g.seg_row0[q] && n < g.seg_row0[q] + g.seg_rows[q]) sg = q;
    if (sg >= 0) {
      float* dst = g.seg_dw[sg] + (long)(n - g.seg_row0[sg]) * g.K_out + k;
      const float v = acc * g.out_scale;
      *dst = (g.beta != 0.f) ? (*dst * g.beta + v) : v;
    }
  }
}

template <int TM, int TN, int NPASS, bool DYB, bool XB>
int launch_tn(const hftt_gemm_tn_desc& d, const TnPlan& p, hipStream_t st) {
  using Cfg = TnCfg<TM, TN, NPASS, DYB, XB>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<TM, TN, NPASS, DYB, XB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) { hftt_set_error("gemm_tn: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  const int tiles = p.n_tiles * p.k_tiles;
  dim3 grid((unsigned)(((p.splits + 7) / 8) * 8 * tiles), 1, 1);
  hipLaunchKernelGGL((gemm_tn_kernel<TM, TN, NPASS, DYB, XB>), grid, dim3(512), Cfg::LDS_BYTES, st, d, p.k_tiles, p.rows_per_split, p.nws, p.kws,
                     tiles, p.splits);
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
  {
    const int ye = (d->io_flags & HFTT_TN_DY_BF16) ? 8 : 4, xe = (d->io_flags & HFTT_TN_X_BF16) ? 8 : 4;
    HFTT_REQUIRE(d->N % ye == 0 && d->K % xe == 0, "gemm_tn: N=%d / K=%d must be multiples of %d / %d", d->N, d->K, ye, xe);
    HFTT_REQUIRE(d->lddy % ye == 0 && d->ldx % xe == 0, "gemm_tn: leading dims must keep rows 16-byte aligned");
  }
  HFTT_REQUIRE(((uintptr_t)d->dY & 15) == 0 && ((uintptr_t)d->X & 15) == 0, "gemm_tn: dY/X must be 16-byte aligned");
  HFTT_REQUIRE(d->npass >= 1 && d->npass <= 4, "gemm_tn: npass must be 1 .. 4");
  HFTT_REQUIRE(d->io_flags == 0 || d->npass == 1 || (d->npass == 4 && (d->io_flags & 3u) != (HFTT_TN_DY_BF16 | HFTT_TN_X_BF16)),
               "gemm_tn: bf16-stored operands need npass == 1, or npass == 4 with ONE of them");
  HFTT_REQUIRE(!(d->io_flags & HFTT_TN_DY_HI) || d->npass == 4, "gemm_tn: HFTT_TN_DY_HI goes with npass == 4");
  HFTT_REQUIRE(d->n_seg >= 1 && d->n_seg <= 8, "gemm_tn: n_seg must be 1..8");
  HFTT_REQUIRE(d->K_out > 0 && d->K_out <= d->K, "gemm_tn: K_out out of range");
  for (int s = 0; s < d->n_seg; s++) {
    HFTT_REQUIRE(d->seg_dw[s] != nullptr, "gemm_tn: null segment pointer");
    HFTT_REQUIRE(d->seg_row0[s] >= 0 && d->seg_rows[s] > 0 && d->seg_row0[s] + d->seg_rows[s] <= d->N, "gemm_tn: segment %d out of range", s);
  }
  HFTT_REQUIRE(d->ws != nullptr && d->ws_bytes >= hftt_gemm_tn_ws_bytes(d->M, d->N, d->K), "gemm_tn: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const TnPlan p = tn_plan(d->M, d->N, d->K);
  int rc;
  const bool dyb = d->io_flags & HFTT_TN_DY_BF16, xb = d->io_flags & HFTT_TN_X_BF16;
  if (d->npass == 3) {
    if (p.tm == 2) rc = launch_tn<2, 4, 3, false, false>(*d, p, st);
    else if (p.tn == 4) rc = launch_tn<1, 4, 3, false, false>(*d, p, st);
    else if (p.tn == 2) rc = launch_tn<1, 2, 3, false, false>(*d, p, st);
    else rc = launch_tn<1, 1, 3, false, false>(*d, p, st);
  } else if (d->npass == 4 && (d->io_flags & HFTT_TN_DY_HI) && !dyb) {
#ifndef HFTT_GRAD_HI_BUILD
    hftt_set_error("gemm_tn: this library was built without the gradient-rounding option (HFTT_BUILD_GRAD_HI=1 python nylon-amt_amd/build.py)");
    return 1;
#else
#define HFTT_TN_GO5(TM_, TN_) (xb ? launch_tn<TM_, TN_, 5, false, true>(*d, p, st) : launch_tn<TM_, TN_, 5, false, false>(*d, p, st))
    if (p.tm == 2) rc = HFTT_TN_GO5(2, 4);
    else if (p.tn == 4) rc = HFTT_TN_GO5(1, 4);
    else if (p.tn == 2) rc = HFTT_TN_GO5(1, 2);
    else rc = HFTT_TN_GO5(1, 1);
#undef HFTT_TN_GO5
#endif
  } else if (d->npass == 4 && (d->io_flags & HFTT_TN_DY_DROP)) {
    HFTT_REQUIRE(!dyb && d->lddy == d->N && d->drop_p >= 0.f && d->drop_p < 1.f, "gemm_tn: HFTT_TN_DY_DROP takes an fp32 dY with lddy == N and 0 <= drop_p < 1");
    HFTT_REQUIRE(d->M < (1 << 24) && d->N % 4 == 0 && (((int64_t)d->M * d->N) >> 2) < (1ll << 32), "gemm_tn: HFTT_TN_DY_DROP indexes the site with 32-bit quads (M < 2^24, M * N < 2^34)");
    if (p.tm == 2) rc = xb ? launch_tn<2, 4, 6, false, true>(*d, p, st) : launch_tn<2, 4, 6, false, false>(*d, p, st);
    else if (p.tn == 4) rc = xb ? launch_tn<1, 4, 6, false, true>(*d, p, st) : launch_tn<1, 4, 6, false, false>(*d, p, st);
    else { hftt_set_error("gemm_tn: HFTT_TN_DY_DROP covers N >= 256 and K >= 256 (got N=%d K=%d)", d->N, d->K); return 1; }
  } else if (d->npass == 4) {
#define HFTT_TN_GO4(TM_, TN_)                                                     \
    (dyb ? launch_tn<TM_, TN_, 4, true, false>(*d, p, st) : (xb ? launch_tn<TM_, TN_, 4, false, true>(*d, p, st) : launch_tn<TM_, TN_, 4, false, false>(*d, p, st)))
    if (p.tm == 2) rc = HFTT_TN_GO4(2, 4);
    else if (p.tn == 4) rc = HFTT_TN_GO4(1, 4);
    else if (p.tn == 2) rc = HFTT_TN_GO4(1, 2);
    else rc = HFTT_TN_GO4(1, 1);
#undef HFTT_TN_GO4
  } else if (d->npass == 2) {
    if (p.tm == 2) rc = launch_tn<2, 4, 2, false, false>(*d, p, st);
    else if (p.tn == 4) rc = launch_tn<1, 4, 2, false, false>(*d, p, st);
    else if (p.tn == 2) rc = launch_tn<1, 2, 2, false, false>(*d, p, st);
    else rc = launch_tn<1, 1, 2, false, false>(*d, p, st);
  } else {
#define HFTT_TN_GO(TM_, TN_)                                                     \
    (dyb ? (xb ? launch_tn<TM_, TN_, 1, true, true>(*d, p, st) : launch_tn<TM_, TN_, 1, true, false>(*d, p, st)) \
         : (xb ? launch_tn<TM_, TN_, 1, false, true>(*d, p, st) : launch_tn<TM_, TN_, 1, false, false>(*d, p, st)))
    if (p.tm == 2) rc = HFTT_TN_GO(2, 4);
    else if (p.tn == 4) rc = HFTT_TN_GO(1, 4);
    else if (p.tn == 2) rc = HFTT_TN_GO(1, 2);
    else rc = HFTT_TN_GO(1, 1);
#undef HFTT_TN_GO
  }
  if (rc != 0) return rc;
  const long total = (long)d->N * d->K_out;
  const int bias_blocks = (d->N + 3) / 4;
  if (total <= 65536 && p.splits >= 32) {             // small output, many partial tiles: the wide reduce
    const int blocks = (int)((total + 15) / 16);
    hipLaunchKernelGGL(gemm_tn_reduce_small_kernel, dim3(blocks + bias_blocks), dim3(256), 0, st, *d, p.splits, p.nws, p.kws, blocks);
    HFTT_CHECK_LAUNCH("gemm_tn_reduce");
    return 0;
  }
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks + bias_blocks), dim3(256), 0, st, *d, p.splits, p.nws, p.kws, blocks);
  HFTT_CHECK_LAUNCH("gemm_tn_reduce");
  return 0;
}
