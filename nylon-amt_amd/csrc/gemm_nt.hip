// NT GEMM with fused epilogue on MFMA (gfx950).
//   C[M,N] = epi(A[M,K] . W[N,K]^T + bias)        -- see include/hftt_hip.h (hftt_gemm_nt)
// Tile: BM=128 x BN in {64,128,256} x BK=32, 512 threads (8 waves, 2 per SIMD), double-buffered LDS,
// register-staged global->LDS.
//   npass == 1 ("bf16"):  fp32 activations are rounded to bf16 on the way into LDS, weights come as a prepared bf16
//                         plane; v_mfma_f32_32x32x16_bf16; LDS rows padded to 80 B (conflict-free ds_read_b128).
//   npass == 3 ("parity"): operands stay fp32 in LDS (rows of 33 floats: conflict-free ds_read_b32), weights come as a
//                         prepared fp32 matrix; v_mfma_f32_32x32x2_f32 = exact fp32 FMA chains (<= 1e-3 parity mode).
#include <stdlib.h>
#include <type_traits>
#include "hftt_common.h"
#include "x3_common.h"
#include "hftt_host.h"
#include "../../include/hftt_hip.h"

namespace {

// gate value > 0 for an fp32 or a bf16 gate tensor (4 consecutive elements at element offset `off`)
__device__ __forceinline__ float4 load_gate4(const float* gate, bool bf, long off) {
  if (bf) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(gate) + off);
    return make_float4(bf2f(u.x & 0xFFFFu), bf2f(u.x >> 16), bf2f(u.y & 0xFFFFu), bf2f(u.y >> 16));
  }
  return *reinterpret_cast<const float4*>(gate + off);
}
// store 4 consecutive outputs as fp32 (16 B) or bf16 (8 B)
__device__ __forceinline__ void store_c4(float* C, bool bf, long off, float a, float b, float c, float d) {
  if (bf) {
    uint2 u;
    u.x = f2bf(a) | ((unsigned)f2bf(b) << 16); u.y = f2bf(c) | ((unsigned)f2bf(d) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(C) + off) = u;
  } else {
    *reinterpret_cast<float4*>(C + off) = make_float4(a, b, c, d);
  }
}

constexpr int BM = 128;
constexpr int BK = 32;

// PREC: 1 = bf16 operands, 3 = exact fp32 MFMA, 2 / 4 = split operands in three passes (x3_common.h: fp16 / bf16 halves; the A tile
// is split on its way into LDS, the weights arrive as two prepared 16-bit planes W / W_lo)
template <int BN, int PREC>
struct NtCfg {
  static constexpr bool F32 = (PREC == 3);
  static constexpr bool X3M = (PREC == 2 || PREC == 4 || PREC == 5);   // 5 = npass 4 with HFTT_NT_A_HI: A (a gradient) enters as its bf16 rounding
  static constexpr int WM = (BN == 64) ? 4 : 2;
  static constexpr int WN = 8 / WM;
  static constexpr int TM = BM / WM / 32;
  static constexpr int TN = BN / WN / 32;
  static constexpr int RS = F32 ? 33 : 40;                 // row stride in elements (floats / bf16)
  static constexpr int ESZ = F32 ? 4 : 2;
  static constexpr int PLANES = X3M ? 2 : 1;               // x3: hi plane, then lo plane
  static constexpr int A_ELEMS = PLANES * BM * RS;
  static constexpr int W_ELEMS = PLANES * BN * RS;
  static constexpr int BUF_ELEMS = A_ELEMS + W_ELEMS;
  static constexpr int LOOP_BYTES = 2 * BUF_ELEMS * ESZ;
  static constexpr int STAGE_LD = BN + 4;
  static constexpr int STAGE_BYTES = BM * STAGE_LD * 4;
  static constexpr int WCHP = (BN * 4 + 511) / 512;        // 16-byte chunks per thread of one 16-bit weight plane
  static constexpr int WCH = F32 ? (BN * 8 + 511) / 512 : PLANES * WCHP;   // 16-byte weight chunks per thread
};

template <int BN, int PREC, bool LN>
__global__ __launch_bounds__(512) void gemm_nt_kernel(const hftt_gemm_nt_desc g) {
  using Cfg = NtCfg<BN, PREC>;
  constexpr bool F32 = Cfg::F32, X3M = Cfg::X3M;
  constexpr int XE = X3M ? (PREC == 5 ? 4 : PREC) : X3_BF16;          // element type of the split (unused otherwise)
  constexpr bool AH = (PREC == 5);
  constexpr int WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN, RS = Cfg::RS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* sm16 = reinterpret_cast<unsigned short*>(smem);
  float* sm32 = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int nk = g.K / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  float4 areg[2];
  uint4 abreg = make_uint4(0u, 0u, 0u, 0u);        // A tile chunk when A is stored as bf16 (16 B = 8 elements per thread)
  const bool a_bf = !F32 && !X3M && (g.io_flags & HFTT_NT_A_BF16);
  const bool c_bf = !F32 && !X3M && (g.io_flags & HFTT_NT_C_BF16);
  const bool gate_bf = !F32 && !X3M && (g.io_flags & HFTT_NT_GATE_BF16);
  uint4 wreg[Cfg::WCH];
#pragma unroll
  for (int j = 0; j < Cfg::WCH; j++) wreg[j] = make_uint4(0u, 0u, 0u, 0u);
  const float* Wf = reinterpret_cast<const float*>(g.W);
  const unsigned short* Wb = reinterpret_cast<const unsigned short*>(g.W);
  const unsigned short* Wlo = reinterpret_cast<const unsigned short*>(g.W_lo);

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    if (a_bf) {
      const int row = tid >> 2, ch = tid & 3;
      const long grow = m0 + row;
      const long gr = grow < g.M ? grow : (long)g.M - 1;
      const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(g.A) + gr * g.lda + k0 + ch * 8);
      abreg = grow < g.M ? t : make_uint4(0u, 0u, 0u, 0u);
    } else
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = tid + 512 * j;
      const int row = i >> 3, c4 = i & 7;
      const long grow = m0 + row;
      // unconditional load from a clamped row + select (a branch around the load makes hipcc serialise the loads)
      const long gr = grow < g.M ? grow : (long)g.M - 1;
      const float4 t = *reinterpret_cast<const float4*>(g.A + gr * g.lda + k0 + c4 * 4);
      areg[j] = grow < g.M ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < Cfg::WCH; j++) {
      const int i = tid + 512 * j;
      if (F32) {
        if (i < BN * 8) {
          const int row = i >> 3, ch = i & 7;
          wreg[j] = *reinterpret_cast<const uint4*>(Wf + (long)(n0 + row) * g.K + k0 + ch * 4);
        }
      } else if (X3M) {
        const int pl = j / Cfg::WCHP, ip = tid + 512 * (j % Cfg::WCHP);
        if (ip < BN * 4) {
          const int row = ip >> 2, ch = ip & 3;
          wreg[j] = *reinterpret_cast<const uint4*>((pl ? Wlo : Wb) + (long)(n0 + row) * g.K + k0 + ch * 8);
        }
      } else {
        if (i < BN * 4) {
          const int row = i >> 2, ch = i & 3;
          wreg[j] = *reinterpret_cast<const uint4*>(Wb + (long)(n0 + row) * g.K + k0 + ch * 8);
        }
      }
    }
  };
  auto sstore = [&](int buf) {
    if (F32) {
      float* As = sm32 + buf * Cfg::BUF_ELEMS;
      float* Ws = As + Cfg::A_ELEMS;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int i = tid + 512 * j;
        const int row = i >> 3, c4 = i & 7;
        float* d = As + row * RS + c4 * 4;
        d[0] = areg[j].x; d[1] = areg[j].y; d[2] = areg[j].z; d[3] = areg[j].w;
      }
#pragma unroll
      for (int j = 0; j < Cfg::WCH; j++) {
        const int i = tid + 512 * j;
        if (i < BN * 8) {
          const int row = i >> 3, ch = i & 7;
          float* d = Ws + row * RS + ch * 4;
          d[0] = __uint_as_float(wreg[j].x); d[1] = __uint_as_float(wreg[j].y);
          d[2] = __uint_as_float(wreg[j].z); d[3] = __uint_as_float(wreg[j].w);
        }
      }
    } else if (X3M) {
      unsigned short* As = sm16 + buf * Cfg::BUF_ELEMS;
      unsigned short* Ws = As + Cfg::A_ELEMS;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int i = tid + 512 * j;
        const int row = i >> 3, c4 = i & 7;
        uint2 hi, lo;
        x3_split4<XE>(areg[j], hi, lo);
        *reinterpret_cast<uint2*>(As + row * RS + c4 * 4) = hi;
        if (!AH) *reinterpret_cast<uint2*>(As + BM * RS + row * RS + c4 * 4) = lo;
      }
#pragma unroll
      for (int j = 0; j < Cfg::WCH; j++) {
        const int pl = j / Cfg::WCHP, ip = tid + 512 * (j % Cfg::WCHP);
        if (ip < BN * 4) {
          const int row = ip >> 2, ch = ip & 3;
          *reinterpret_cast<uint4*>(Ws + pl * BN * RS + row * RS + ch * 8) = wreg[j];
        }
      }
    } else {
      unsigned short* As = sm16 + buf * Cfg::BUF_ELEMS;
      unsigned short* Ws = As + Cfg::A_ELEMS;
      if (a_bf) {
        *reinterpret_cast<uint4*>(As + (tid >> 2) * RS + (tid & 3) * 8) = abreg;
      } else
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int i = tid + 512 * j;
        const int row = i >> 3, c4 = i & 7;
        uint2 ph;
        ph.x = f2bf(areg[j].x) | ((unsigned)f2bf(areg[j].y) << 16);
        ph.y = f2bf(areg[j].z) | ((unsigned)f2bf(areg[j].w) << 16);
        *reinterpret_cast<uint2*>(As + row * RS + c4 * 4) = ph;
      }
#pragma unroll
      for (int j = 0; j < Cfg::WCH; j++) {
        const int i = tid + 512 * j;
        if (i < BN * 4) {
          const int row = i >> 2, ch = i & 3;
          *reinterpret_cast<uint4*>(Ws + row * RS + ch * 8) = wreg[j];
        }
      }
    }
  };

  gload(0);
  sstore(0);
  __syncthreads();

  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    if (F32) {
      const float* As = sm32 + buf * Cfg::BUF_ELEMS;
      const float* Ws = As + Cfg::A_ELEMS;
      // Two-level summation: the 32 k values of a stage are accumulated from zero (16 MFMA steps), then added to the running sum -- a
      // chain of 16 + K/32 roundings instead of K/2.  One running accumulator over K = 256 measured 1.35x (rms) / 2x (max) the error of a
      // CPU sgemm on the first encoder layer's Q / K projections (|x| ~ 600), and those feed logits of ~1e5 whose near-ties decide the
      // first layer's gradients (tests/dev_layer_diff.py).
      f32x16 part[TM][TN];
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) part[i][j][r] = 0.f;
#pragma unroll
      for (int t = 0; t < 16; t++) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) a[i] = As[(wm * TM * 32 + i * 32 + lr) * RS + 16 * lh + t];
#pragma unroll
        for (int j = 0; j < TN; j++) b[j] = Ws[(wn * TN * 32 + j * 32 + lr) * RS + 16 * lh + t];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) part[i][j] = mfma32_f32(a[i], b[j], part[i][j]);
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] += part[i][j];
    } else if (X3M) {
      const unsigned short* As = sm16 + buf * Cfg::BUF_ELEMS;
      const unsigned short* Ws = As + Cfg::A_ELEMS;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) {
          const unsigned short* p = As + (wm * TM * 32 + i * 32 + lr) * RS + s * 16 + lh * 8;
          ah[i] = lds_read_b128(p);
          if (!AH) al[i] = lds_read_b128(p + BM * RS);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
          const unsigned short* p = Ws + (wn * TN * 32 + j * 32 + lr) * RS + s * 16 + lh * 8;
          bh[j] = lds_read_b128(p); bl[j] = lds_read_b128(p + BN * RS);
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) {
            if (AH) { acc[i][j] = X3<XE>::mma(ah[i], bl[j], acc[i][j]); acc[i][j] = X3<XE>::mma(ah[i], bh[j], acc[i][j]); }
            else acc[i][j] = x3_mma<XE>(ah[i], al[i], bh[j], bl[j], acc[i][j]);
          }
      }
    } else {
      const unsigned short* As = sm16 + buf * Cfg::BUF_ELEMS;
      const unsigned short* Ws = As + Cfg::A_ELEMS;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) a[i] = lds_read_b128(As + (wm * TM * 32 + i * 32 + lr) * RS + s * 16 + lh * 8);
#pragma unroll
        for (int j = 0; j < TN; j++) b[j] = lds_read_b128(Ws + (wn * TN * 32 + j * 32 + lr) * RS + s * 16 + lh * 8);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
      }
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ------------------------------- epilogue -------------------------------
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  float* stage = reinterpret_cast<float*>(smem);
  // Split modes, full 256-column tiles: the fp32 result leaves through LDS as whole rows (one wave per row, 16 bytes per lane) and the
  // table add / gate / dropout / residual run there per QUAD of columns.  From the accumulator layout every element paid its own
  // `row % add_mod`, its own hash and a 4-byte store: the embedding GEMM (K = 96, all epilogue) took 440 us for 268 MB at S_e.
  const bool rowpass = X3M && !LN && BN == 256 && (g.N % 4 == 0) && (g.ldc % 4 == 0) && (g.gate == nullptr || g.ldg % 4 == 0) &&
                       (g.residual == nullptr || g.ldr % 4 == 0) && ((((uintptr_t)g.C | (uintptr_t)g.gate | (uintptr_t)g.residual | (uintptr_t)g.add_table) & 15) == 0);

#pragma unroll
  for (int i = 0; i < TM; i++) {
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col_l = wn * TN * 32 + j * 32 + lr;
      const int col = n0 + col_l;
      const bool col_ok = col < g.N;
      const float bv = (g.bias != nullptr && col_ok) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row_l = wm * TM * 32 + i * 32 + acc_row32(r, lh);
        const long row = m0 + row_l;
        const bool ok = col_ok && row < g.M;
        float v = acc[i][j][r] + bv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        v *= g.out_scale;
        if (rowpass) { stage[row_l * Cfg::STAGE_LD + col_l] = v; continue; }      // (wave-uniform)
        if (ok) {
          if (g.add_table != nullptr) v += g.add_table[(long)(row % g.add_mod) * g.N + col];
          if (g.gate != nullptr) {
            const float gv = gate_bf ? bf2f(reinterpret_cast<const unsigned short*>(g.gate)[row * g.ldg + col]) : g.gate[row * g.ldg + col];
            v = (gv > 0.f) ? v * g.gate_scale : 0.f;
          }
          if (g.drop_p > 0.f) v = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)row * g.N + col, thr) ? v * inv_keep : 0.f;
          if (g.residual != nullptr) v += g.residual[(long)(row % g.res_mod) * g.ldr + col];
        }
        if (LN) {
          stage[row_l * Cfg::STAGE_LD + col_l] = v;
        } else if (ok) {
          if (c_bf) reinterpret_cast<unsigned short*>(g.C)[row * g.ldc + col] = f2bf(v);
          else g.C[row * g.ldc + col] = v;
        }
      }
    }
  }

  if (rowpass) {
    __syncthreads();
    const int col = n0 + lane * 4;
    for (int rr = 0; rr < BM / 8; rr++) {
      const int row_l = wave * (BM / 8) + rr;
      const long row = m0 + row_l;
      if (row >= g.M) break;   // wave-uniform
      if (col >= g.N) continue;
      float4 v = *reinterpret_cast<const float4*>(stage + row_l * Cfg::STAGE_LD + lane * 4);
      if (g.add_table != nullptr) {
        const float4 t = *reinterpret_cast<const float4*>(g.add_table + (long)(row % g.add_mod) * g.N + col);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (g.gate != nullptr) {
        const float4 gv = *reinterpret_cast<const float4*>(g.gate + row * g.ldg + col);
        v.x = gv.x > 0.f ? v.x * g.gate_scale : 0.f; v.y = gv.y > 0.f ? v.y * g.gate_scale : 0.f;
        v.z = gv.z > 0.f ? v.z * g.gate_scale : 0.f; v.w = gv.w > 0.f ? v.w * g.gate_scale : 0.f;
      }
      if (g.drop_p > 0.f) {
        const uint32_t k4 = hftt_keep_quad(g.drop_seed, g.drop_site, ((uint64_t)row * (uint64_t)g.N + (uint64_t)col) >> 2, thr);
        v.x = (k4 & 1u) ? v.x * inv_keep : 0.f; v.y = (k4 & 2u) ? v.y * inv_keep : 0.f;
        v.z = (k4 & 4u) ? v.z * inv_keep : 0.f; v.w = (k4 & 8u) ? v.w * inv_keep : 0.f;
      }
      if (g.residual != nullptr) {
        const float4 t = *reinterpret_cast<const float4*>(g.residual + (long)(row % g.res_mod) * g.ldr + col);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      *reinterpret_cast<float4*>(g.C + row * g.ldc + col) = v;
    }
  }

  if (LN) {
    __syncthreads();
    constexpr int VPL = BN / 64;
    for (int rr = 0; rr < BM / 8; rr++) {
      const int row_l = wave * (BM / 8) + rr;
      const long row = m0 + row_l;
      if (row >= g.M) break;   // wave-uniform
      float v[VPL];
#pragma unroll
      for (int e = 0; e < VPL; e++) v[e] = stage[row_l * Cfg::STAGE_LD + lane * VPL + e];
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < VPL; e++) s += v[e];
      const float mean = wave_sum(s) * (1.0f / BN);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < VPL; e++) { const float dlt = v[e] - mean; q += dlt * dlt; }
      const float var = wave_sum(q) * (1.0f / BN);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
      for (int e = 0; e < VPL; e++) {
        const int col = lane * VPL + e;
        if (g.pre_ln_out != nullptr) g.pre_ln_out[row * g.ldc + col] = v[e];
        g.C[row * g.ldc + col] = (v[e] - mean) * rstd * g.ln_gamma[col] + g.ln_beta[col];
      }
      if (lane == 0) {
        if (g.ln_mean != nullptr) g.ln_mean[row] = mean;
        if (g.ln_rstd != nullptr) g.ln_rstd[row] = rstd;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 mode, A-stationary persistent form (N a multiple of 256, K <= 768).  These GEMMs are tall and short-K
// (M ~ 10^5 tokens): the kernel is a streaming kernel with MFMA work hidden underneath.
//   * persistent workgroups (2 per CU) walk the 32-row blocks of A;
//   * a block of A (32 x K fp32) is read from HBM ONCE, converted to bf16 and kept in LDS; with DBUF the NEXT block's loads
//     are issued before the current block's MFMA loop and land in the other LDS buffer afterwards (HBM latency hidden);
//   * the (L2-resident) weight tiles of the current 256-wide N tile stream through a double-buffered LDS ring shared by the
//     8 waves (a per-wave L2->register stream was measured 30 % slower: twice the L2 traffic in 32-byte pieces);
//   * every N tile is finished through LDS: accumulators -> stage (the weight ring's space) -> one wave per row, 16 B per
//     lane: table add, gate, dropout, residual, LayerNorm, float4 stores.  (Direct 4-byte stores from the accumulator
//     layout were measured at 2.8 TB/s; the staged float4 form reaches > 4 TB/s.)
// A is read once per GEMM regardless of N.
// ------------------------------------------------------------------------------------------------------------------
template <int APF, bool DBUF, bool ABF>
__global__ __launch_bounds__(512, (DBUF ? 2 : 4)) void gemm_nt_as_kernel(const hftt_gemm_nt_desc g) {
  constexpr int BM_ = 32;
  constexpr int BN = 256;
  constexpr int RSW = 40;                          // W tile row stride (bf16): 80 B
  constexpr int W_ELEMS = BN * RSW;
  constexpr int WCH = 2;                           // 16-byte chunks per thread per W tile (256 rows x 4 chunks / 512)
  constexpr int STAGE_LD = BN + 4;
  static_assert(BM_ * STAGE_LD * 4 <= 2 * W_ELEMS * 2, "stage must fit the weight ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = g.K;
  const int RSA = K + 8;                           // A row stride (bf16): 2K+16 bytes, conflict-free ds_read_b128
  const int A_ELEMS = BM_ * RSA;
  unsigned short* As0 = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Ws = As0 + (DBUF ? 2 : 1) * A_ELEMS;
  float* stage = reinterpret_cast<float*>(Ws);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int KT = K / 32;
  const int NT = (g.N + BN - 1) / BN;
  const int steps = NT * KT;
  const int f4r = K >> 2;
  const int a_total = BM_ * f4r;
  const long nblk = (g.M + BM_ - 1) / BM_;
  const unsigned short* Wb = reinterpret_cast<const unsigned short*>(g.W);
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);

  uint4 wreg[WCH] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  auto wload = [&](int s) {
    const int nt = s / KT, kt = s - nt * KT;
#pragma unroll
    for (int j = 0; j < WCH; j++) {
      const int i = tid + 512 * j;
      const int row = i >> 2, ch = i & 3;
      wreg[j] = *reinterpret_cast<const uint4*>(Wb + (long)(nt * BN + row) * K + kt * 32 + ch * 8);
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < WCH; j++) {
      const int i = tid + 512 * j;
      const int row = i >> 2, ch = i & 3;
      *reinterpret_cast<uint4*>(Ws + buf * W_ELEMS + row * RSW + ch * 8) = wreg[j];
    }
  };
  constexpr bool a_bf = ABF;                       // A stored as bf16: APF counts 16-byte chunks either way
  const bool c_bf = g.io_flags & HFTT_NT_C_BF16, gate_bf = g.io_flags & HFTT_NT_GATE_BF16, res_bf = g.io_flags & HFTT_NT_RES_BF16;
  const int c8r = K >> 3;                          // 16-byte chunks per row when A is stored as bf16
  const int a_total8 = BM_ * c8r;
  float4 apf[APF];
  auto aload = [&](long blk) {                     // HBM -> registers (APF x 16 B in flight per thread)
    if (a_bf) {
#pragma unroll
      for (int u = 0; u < APF; u++) {
        const int i = tid + 512 * u;
        const int ic = i < a_total8 ? i : a_total8 - 1;
        const int row = ic / c8r, c8 = ic - row * c8r;
        const long grow = blk * BM_ + row;
        const long gr = grow < g.M ? grow : (long)g.M - 1;
        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned short*>(g.A) + gr * g.lda + c8 * 8);
        apf[u] = (i < a_total8 && grow < g.M) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < APF; u++) {
      const int i = tid + 512 * u;
      const int ic = i < a_total ? i : a_total - 1;        // clamped: loads stay unconditional
      const int row = ic / f4r, c4 = ic - row * f4r;
      const long grow = blk * BM_ + row;
      const long gr = grow < g.M ? grow : (long)g.M - 1;
      const float4 t = *reinterpret_cast<const float4*>(g.A + gr * g.lda + c4 * 4);
      apf[u] = (i < a_total && grow < g.M) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto astore = [&](unsigned short* As) {          // registers -> bf16 -> LDS
    if (a_bf) {
#pragma unroll
      for (int u = 0; u < APF; u++) {
        const int i = tid + 512 * u;
        if (i < a_total8) {
          const int row = i / c8r, c8 = i - row * c8r;
          *reinterpret_cast<float4*>(As + row * RSA + c8 * 8) = apf[u];
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < APF; u++) {
      const int i = tid + 512 * u;
      if (i < a_total) {
        const int row = i / f4r, c4 = i - row * f4r;
        uint2 ph;
        ph.x = f2bf(apf[u].x) | ((unsigned)f2bf(apf[u].y) << 16);
        ph.y = f2bf(apf[u].z) | ((unsigned)f2bf(apf[u].w) << 16);
        *reinterpret_cast<uint2*>(As + row * RSA + c4 * 4) = ph;
      }
    }
  };

  long blk = blockIdx.x;
  if (blk >= nblk) return;
  wload(0);
  aload(blk);
  astore(As0);
  wstore(0);
  __syncthreads();

  int cur = 0;
  for (; blk < nblk; blk += gridDim.x) {
    const unsigned short* As = As0 + (DBUF ? cur : 0) * A_ELEMS;
    const long m0 = blk * BM_;
    const long nxt = blk + gridDim.x;
    if (DBUF && nxt < nblk) aload(nxt);            // in flight during the whole MFMA loop of this block

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    int nt = 0, kt = 0;
    for (int s = 0; s < steps; s++) {
      const int buf = s & 1;
      const bool last = (s + 1 == steps);
      if (!last) wload(s + 1);
      else if (nxt < nblk) wload(0);               // first weight tile of the next block
      const unsigned short* Wt = Ws + buf * W_ELEMS;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const bf16x8 a = lds_read_b128(As + lr * RSA + kt * 32 + ks * 16 + lh * 8);
        const bf16x8 b = lds_read_b128(Wt + (wave * 32 + lr) * RSW + ks * 16 + lh * 8);
        acc = mfma32(a, b, acc);
      }
      if (kt != KT - 1) {
        wstore(buf ^ 1);
        kt++;
      } else {
        // ---------------- epilogue of N tile nt: accumulators -> stage (weight ring) -> row pass ----------------
        const int n0 = nt * BN;
        __syncthreads();                             // all MFMA reads of the weight ring are done
        {
          const int col_l = wave * 32 + lr;
          const int col = n0 + col_l;
          const float bv = (g.bias != nullptr && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            float v = acc[r] + bv;
            acc[r] = 0.f;
            if (g.act == 1) v = fmaxf(v, 0.f);
            stage[acc_row32(r, lh) * STAGE_LD + col_l] = v * g.out_scale;
          }
        }
        __syncthreads();
        {
          constexpr int RPW = BM_ / 8;               // 4 rows per wave, finished two at a time
          constexpr int RB = 2;
          const int c4 = n0 + lane * 4;
          float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g.ln_gamma != nullptr) {
            ga = *reinterpret_cast<const float4*>(g.ln_gamma + c4);
            be = *reinterpret_cast<const float4*>(g.ln_beta + c4);
          }
#pragma unroll 1
          for (int rb = 0; rb < RPW; rb += RB) {
          float4 tab[RB], gat[RB], res[RB];
#pragma unroll
          for (int rr = 0; rr < RB; rr++) {
            const long row = m0 + wave * RPW + rb + rr;
            tab[rr] = make_float4(0.f, 0.f, 0.f, 0.f); gat[rr] = tab[rr]; res[rr] = tab[rr];
            const long rc = row < g.M ? row : (long)g.M - 1;      // clamped row: the (wave-uniform) pointer tests are the only branches
            if (g.add_table != nullptr) tab[rr] = *reinterpret_cast<const float4*>(g.add_table + (long)(rc % g.add_mod) * g.N + c4);
            if (g.gate != nullptr) gat[rr] = load_gate4(g.gate, gate_bf, rc * g.ldg + c4);
            if (g.residual != nullptr) res[rr] = load_gate4(g.residual, res_bf, (long)(rc % g.res_mod) * g.ldr + c4);
          }
#pragma unroll
          for (int rr = 0; rr < RB; rr++) {
            const int row_l = wave * RPW + rb + rr;
            const long row = m0 + row_l;
            if (row < g.M) {         // wave-uniform
              const float4 sv = *reinterpret_cast<const float4*>(stage + row_l * STAGE_LD + lane * 4);
              float v[4] = {sv.x + tab[rr].x, sv.y + tab[rr].y, sv.z + tab[rr].z, sv.w + tab[rr].w};
              if (g.gate != nullptr) {
                v[0] = gat[rr].x > 0.f ? v[0] * g.gate_scale : 0.f; v[1] = gat[rr].y > 0.f ? v[1] * g.gate_scale : 0.f;
                v[2] = gat[rr].z > 0.f ? v[2] * g.gate_scale : 0.f; v[3] = gat[rr].w > 0.f ? v[3] * g.gate_scale : 0.f;
              }
              if (g.drop_p > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; e++)
                  v[e] = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)row * g.N + c4 + e, thr) ? v[e] * inv_keep : 0.f;
              }
              v[0] += res[rr].x; v[1] += res[rr].y; v[2] += res[rr].z; v[3] += res[rr].w;
              if (g.ln_gamma != nullptr) {              // N == 256: the row is complete
                const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / BN);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 4; e++) { const float dlt = v[e] - mean; q += dlt * dlt; }
                const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / BN) + 1e-5f);
                if (g.pre_ln_out != nullptr) *reinterpret_cast<float4*>(g.pre_ln_out + row * g.ldc + c4) = make_float4(v[0], v[1], v[2], v[3]);
                v[0] = (v[0] - mean) * rstd * ga.x + be.x; v[1] = (v[1] - mean) * rstd * ga.y + be.y;
                v[2] = (v[2] - mean) * rstd * ga.z + be.z; v[3] = (v[3] - mean) * rstd * ga.w + be.w;
                if (lane == 0) {
                  if (g.ln_mean != nullptr) g.ln_mean[row] = mean;
                  if (g.ln_rstd != nullptr) g.ln_rstd[row] = rstd;
                }
              }
              store_c4(g.C, c_bf, row * g.ldc + c4, v[0], v[1], v[2], v[3]);
            }
          }
          }
        }
        __syncthreads();                             // row pass finished reading the stage (= weight ring)
        if (!last || nxt < nblk) wstore(last ? 0 : (buf ^ 1));
        kt = 0;
        nt++;
      }
      __syncthreads();
    }
    // ---- next block of A ----
    if (nxt < nblk) {
      if (DBUF) {
        astore(As0 + (cur ^ 1) * A_ELEMS);
        cur ^= 1;
      } else {
        aload(nxt);
        astore(As0);
      }
      __syncthreads();
    }
  }
}

template <int APF, bool DBUF, bool ABF>
int launch_nt_as(const hftt_gemm_nt_desc& d, hipStream_t st) {
  const int lds = ((DBUF ? 2 : 1) * 32 * (d.K + 8) + 2 * 256 * 40) * 2;
  if (lds > 160 * 1024) { hftt_set_error("gemm_nt: K=%d too large for the A-stationary tile (%d B LDS)", d.K, lds); return 1; }
  static int attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_as_kernel<APF, DBUF, ABF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { hftt_set_error("gemm_nt: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e)); return 2; }
    attr_lds = lds;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { hftt_set_error("gemm_nt: device query failed"); return 2; }
    n_cu = prop.multiProcessorCount;
  }
  const long nblk = ((long)d.M + 31) / 32;
  const int per_cu = (lds <= 80 * 1024) ? 2 : 1;
  long grid = (long)n_cu * per_cu;
  if (grid > nblk) grid = nblk;
  hipLaunchKernelGGL((gemm_nt_as_kernel<APF, DBUF, ABF>), dim3((unsigned)grid), dim3(512), lds, st, d);
  HFTT_CHECK_LAUNCH("gemm_nt");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 mode, A-stationary one-shot form for K <= 256 (measured faster there than the persistent form below): one
// workgroup per BM-row block, A block loaded once (all loads in flight), weight tiles through a double-buffered LDS ring.
//   MODE 0  plain (bias / ReLU / scale): stores straight from the accumulators;
//   MODE 1  staged, single N tile (N == 256): tile -> LDS -> one wave per row, 16 B per lane (table add, gate, dropout,
//           residual, LayerNorm with float4 traffic);
//   MODE 2  staged, several N tiles (BM = 32): same row pass per N tile, staged in the weight ring.
// ------------------------------------------------------------------------------------------------------------------
template <int BM_, int MODE, bool EW = false, int PF = 1>
__global__ __launch_bounds__(512, 4) void gemm_nt_as1_kernel(const hftt_gemm_nt_desc g) {
  constexpr int BN = 256;
  constexpr int WM = (BM_ == 32) ? 1 : 2;
  constexpr int WN = 8 / WM;
  constexpr int TN = BN / (32 * WN);
  static_assert(BM_ == 32 * WM, "one 32-row MFMA tile per wave row");
  constexpr int RSW = 40;
  constexpr int W_ELEMS = BN * RSW;
  constexpr int WCH = 2;
  constexpr int STAGE_LD = BN + 4;
  static_assert(MODE != 2 || BM_ * STAGE_LD * 4 <= 2 * W_ELEMS * 2, "stage must fit the weight ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = g.K;
  const int RSA = K + 8;
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Ws = As + BM_ * RSA;
  float* stage = (MODE == 2) ? reinterpret_cast<float*>(Ws) : reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const long m0 = (long)blockIdx.x * BM_;
  const int KT = K / 32;
  const int NT = (g.N + BN - 1) / BN;
  const int steps = NT * KT;
  const unsigned short* Wb = reinterpret_cast<const unsigned short*>(g.W);

  // weight tiles travel global -> registers -> LDS ring; the global load of tile s + PF is issued at step s into one of PF
  // register sets, so an L2 round trip (~700 ns) has PF steps of MFMA work to hide behind
  uint4 wr[PF][WCH];
#pragma unroll
  for (int i = 0; i < PF; i++)
#pragma unroll
    for (int j = 0; j < WCH; j++) wr[i][j] = make_uint4(0u, 0u, 0u, 0u);
  auto wload = [&](uint4 (&w)[WCH], int s) {
    const int nt = s / KT, kt = s - nt * KT;
#pragma unroll
    for (int j = 0; j < WCH; j++) {
      const int i = tid + 512 * j;
      const int row = i >> 2, ch = i & 3;
      w[j] = *reinterpret_cast<const uint4*>(Wb + (long)(nt * BN + row) * K + kt * 32 + ch * 8);
    }
  };
  auto a_off = [&](int row, int c8) { return row * RSA + (c8 << 3); };
  auto w_off = [&](int row, int ch) { return row * RSW + (ch << 3); };
  auto wstore = [&](const uint4 (&w)[WCH], int buf) {
#pragma unroll
    for (int j = 0; j < WCH; j++) {
      const int i = tid + 512 * j;
      const int row = i >> 2, ch = i & 3;
      *reinterpret_cast<uint4*>(Ws + buf * W_ELEMS + w_off(row, ch)) = w[j];
    }
  };
#pragma unroll
  for (int i = 0; i < PF; i++) wload(wr[i], i < steps ? i : steps - 1);
  const bool a_bf = g.io_flags & HFTT_NT_A_BF16, c_bf = g.io_flags & HFTT_NT_C_BF16, gate_bf = g.io_flags & HFTT_NT_GATE_BF16, res_bf = g.io_flags & HFTT_NT_RES_BF16;
  if (a_bf) {                                      // A stored as bf16: 16-byte chunks straight into LDS
    // thread t owns 16-byte chunks t, t + 512, ... of the BM_ x K block (row-major); (row, chunk) advance incrementally --
    // one integer division per thread instead of one per load (the index math was ~20 % of this kernel's VALU time)
    const int c8r = K >> 3;
    const int total = BM_ * c8r;
    const int drow = 512 / c8r, dc8 = 512 - drow * c8r;
    int row = tid / c8r, c8 = tid - row * c8r;
    for (int base = 0; base < total; base += 512 * 8) {
      uint4 v[8];
      int rr[8], cc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        rr[u] = row; cc[u] = c8;
        const bool in = row < BM_;
        const long grow = m0 + row;
        const long gr = (in && grow < g.M) ? grow : (m0 < g.M ? m0 : 0);
        const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(g.A) + gr * g.lda + c8 * 8);
        v[u] = (in && grow < g.M) ? t : make_uint4(0u, 0u, 0u, 0u);
        row += drow; c8 += dc8;
        if (c8 >= c8r) { c8 -= c8r; row++; }
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (rr[u] < BM_) *reinterpret_cast<uint4*>(As + a_off(rr[u], cc[u])) = v[u];
    }
  } else {
    const int f4r = K >> 2;
    const int total = BM_ * f4r;
    const int drow = 512 / f4r, dc4 = 512 - drow * f4r;
    int row = tid / f4r, c4 = tid - row * f4r;
    for (int base = 0; base < total; base += 512 * 8) {
      float4 v[8];
      int rr[8], cc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        rr[u] = row; cc[u] = c4;
        const bool in = row < BM_;                         // clamped address: loads stay unconditional (no branch, no early vmcnt wait)
        const long grow = m0 + row;
        const long gr = (in && grow < g.M) ? grow : (m0 < g.M ? m0 : 0);
        const float4 t = *reinterpret_cast<const float4*>(g.A + gr * g.lda + c4 * 4);
        v[u] = (in && grow < g.M) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        row += drow; c4 += dc4;
        if (c4 >= f4r) { c4 -= f4r; row++; }
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (rr[u] < BM_) {
          typedef __bf16 bf4_t __attribute__((ext_vector_type(4)));
          const f32x4 fv = {v[u].x, v[u].y, v[u].z, v[u].w};
          *reinterpret_cast<uint2*>(As + a_off(rr[u], cc[u] >> 1) + (cc[u] & 1) * 4) = __builtin_bit_cast(uint2, __builtin_convertvector(fv, bf4_t));
        }
    }
  }
  wstore(wr[0], 0);
  __syncthreads();

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);

  // per k step u of an unrolled group of PF (so the register sets are named statically; PF must divide K / 32): set u held tile s
  // (already in the ring) and receives tile s + PF; set u + 1 holds tile s + 1, written to the ring once this step's MFMAs are issued
  constexpr int U = PF;
  int s = 0;
  for (int nt = 0; nt < NT; nt++) {
    for (int kt0 = 0; kt0 < KT; kt0 += U) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int kt = kt0 + u;
        uint4 (&wl)[WCH] = wr[u];
        const uint4 (&wnext)[WCH] = wr[(u + 1) % PF];
        const int buf = (U % 2 == 0) ? (u & 1) : (s & 1);
        // unconditional (index clamped to the last tile): a load inside a conditional block makes hipcc's s_waitcnt insertion
        // fall back to vmcnt(0) at the ring write, which would drain the younger prefetch too
        wload(wl, s + PF < steps ? s + PF : steps - 1);
        const unsigned short* Wt = Ws + buf * W_ELEMS;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          const bf16x8 a = lds_read_b128(As + (wm * 32 + lr) * RSA + kt * 32 + ks * 16 + lh * 8);
#pragma unroll
          for (int j = 0; j < TN; j++) {
            const bf16x8 b = lds_read_b128(Wt + (wn * TN * 32 + j * 32 + lr) * RSW + ks * 16 + lh * 8);
            acc[j] = mfma32(a, b, acc[j]);
          }
        }
        const bool tile_end = (kt == KT - 1);
        if (MODE == 0) wstore(wnext, buf ^ 1);            // past the last tile this rewrites a dead buffer
        else if (!tile_end && s + 1 < steps) wstore(wnext, buf ^ 1);
        if (!tile_end) {
          __syncthreads();
          s++;
        }
      }
    }
    {                                  // N tile finished: s = its last step, tile s + 1 (if any) sits in set 0
      const int buf = s & 1;
      const int n0 = nt * BN;
      if (MODE != 0) __syncthreads();
      if (MODE == 0 && c_bf) {
        // bf16 C straight from the accumulators: lanes 2i / 2i+1 hold adjacent columns and every lane two rows per register pair;
        // pair_rows_to_cols() (cvt_pk + DPP + byte permute) gives each lane one packed pair of adjacent columns = one 4-byte store.
        // Instruction count matters here: the VALU, not HBM, was the busiest unit of this kernel (SQ_ACTIVE_INST_VALU ~ 48 %).
        const bool odd = lane & 1;
        unsigned short* Cb = reinterpret_cast<unsigned short*>(g.C);
        const float lo_clamp = (g.act == 1) ? 0.f : -INFINITY;
        const bool gated = EW && g.gate != nullptr;
        const float oscale = gated ? g.out_scale * g.gate_scale : g.out_scale;
        auto tile_out = [&](auto full_c) {
          constexpr bool FULL = decltype(full_c)::value;       // every row of the block is inside M: no per-store bounds test
#pragma unroll
          for (int j = 0; j < TN; j++) {
            const int col = n0 + wn * TN * 32 + j * 32 + lr;
            const float bv = (g.bias != nullptr) ? g.bias[col] : 0.f;
            unsigned rbase = (unsigned)(wm * 32 + 4 * lh);
            asm volatile("" : "+v"(rbase));       // opaque: keeps the per-row address / index terms from being hoisted out of the k loop
            const uint64_t ebase = (uint64_t)(m0 + rbase) * (unsigned)g.N + (unsigned)col;
            const long row0 = m0 + rbase + (odd ? 1 : 0);
            const long row0c = FULL ? row0 : (row0 < g.M ? row0 : (long)g.M - 1);
            unsigned short* cp = Cb + row0 * g.ldc + (col & ~1);
            const unsigned short* gpt = reinterpret_cast<const unsigned short*>(g.gate) + row0c * g.ldg + (col & ~1);
            unsigned gpk[8];
            if (gated) {                     // all eight gate pairs of this column block in flight before the first one is used
#pragma unroll
              for (int rp = 0; rp < 8; rp++) {
                const int rofs = ((2 * rp) & 3) + 8 * ((2 * rp) >> 2);
                const bool ok = FULL || (row0 + rofs < g.M);
                gpk[rp] = *reinterpret_cast<const unsigned*>(gpt + (ok ? (long)rofs * g.ldg : 0L));
              }
            }
#pragma unroll
            for (int rp = 0; rp < 8; rp++) {
              const int rofs = ((2 * rp) & 3) + 8 * ((2 * rp) >> 2);     // acc_row32(2 * rp, lh) - 4 * lh: the even register's row
              float own0 = fmaxf(acc[j][2 * rp] + bv, lo_clamp) * oscale, own1 = fmaxf(acc[j][2 * rp + 1] + bv, lo_clamp) * oscale;
              acc[j][2 * rp] = 0.f; acc[j][2 * rp + 1] = 0.f;
              if (EW && g.drop_p > 0.f) {    // elementwise extras stay in registers (no staging pass): dropout on the own elements ...
                const uint64_t i0 = ebase + (uint64_t)((unsigned)rofs * (unsigned)g.N);      // element index = row * N + col
                own0 = hftt_keep(g.drop_seed, g.drop_site, i0, thr) ? own0 * inv_keep : 0.f;
                own1 = hftt_keep(g.drop_seed, g.drop_site, i0 + (unsigned)g.N, thr) ? own1 * inv_keep : 0.f;
              }
              unsigned pk = pair_rows_to_cols(own0, own1, odd);
              const bool ok = FULL || (row0 + rofs < g.M);
              if (gated) {                   // ... and the bf16 ReLU gate as one packed pair per lane, same footprint as the store
                const unsigned gp = gpk[rp];
                // keep a half iff its gate is a positive bf16: bits in [0x0001, 0x7FFF]
                const unsigned m = (((gp & 0xFFFFu) - 1u) < 0x7FFFu ? 0x0000FFFFu : 0u) | ((((gp >> 16) - 1u) < 0x7FFFu) ? 0xFFFF0000u : 0u);
                pk &= m;
              }
              if (ok) *reinterpret_cast<unsigned*>(cp + (long)rofs * g.ldc) = pk;
              if (EW) __builtin_amdgcn_sched_barrier(0);      // keep the hash chains of different pairs from overlapping (registers)
            }
          }
        };
        if (m0 + BM_ <= (long)g.M) tile_out(std::true_type{}); else tile_out(std::false_type{});
      } else
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int col_l = wn * TN * 32 + j * 32 + lr;
        const int col = n0 + col_l;
        const float bv = (g.bias != nullptr) ? g.bias[col] : 0.f;       // N % 256 == 0 on this path: every column exists
        const float lo_clamp = (g.act == 1) ? 0.f : -INFINITY;
        int rb0 = wm * 32 + 4 * lh;
        if (MODE == 0) asm volatile("" : "+v"(rb0));     // opaque: the 16 row addresses are formed here, not hoisted above the k loop
        float* cp = g.C + (m0 + rb0) * g.ldc + col;
        float* sp = stage + rb0 * STAGE_LD + col_l;
        const bool full = m0 + BM_ <= (long)g.M;           // block-uniform
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int rofs = (r & 3) + 8 * (r >> 2);           // row = wm * 32 + acc_row32(r, lh) = rb0 + rofs
          const float v = fmaxf(acc[j][r] + bv, lo_clamp) * g.out_scale;
          acc[j][r] = 0.f;
          if (MODE != 0) {
            sp[rofs * STAGE_LD] = v;
          } else {
            if (full || m0 + rb0 + rofs < g.M) cp[(long)rofs * g.ldc] = v;
          }
        }
      }
      if (MODE != 0) {
        __syncthreads();
        constexpr int RPW = BM_ / 8;
        constexpr int RB = RPW < 4 ? RPW : 4;
        const int c4 = n0 + lane * 4;
        float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.ln_gamma != nullptr) {
          ga = *reinterpret_cast<const float4*>(g.ln_gamma + c4);
          be = *reinterpret_cast<const float4*>(g.ln_beta + c4);
        }
#pragma unroll 1
        for (int rb = 0; rb < RPW; rb += RB) {
          float4 tab[RB], gat[RB], res[RB];
#pragma unroll
          for (int rr = 0; rr < RB; rr++) {
            const long row = m0 + wave * RPW + rb + rr;
            tab[rr] = make_float4(0.f, 0.f, 0.f, 0.f); gat[rr] = tab[rr]; res[rr] = tab[rr];
            const long rc = row < g.M ? row : (long)g.M - 1;      // clamped row: the (wave-uniform) pointer tests are the only branches
            if (g.add_table != nullptr) tab[rr] = *reinterpret_cast<const float4*>(g.add_table + (long)(rc % g.add_mod) * g.N + c4);
            if (g.gate != nullptr) gat[rr] = load_gate4(g.gate, gate_bf, rc * g.ldg + c4);
            if (g.residual != nullptr) res[rr] = load_gate4(g.residual, res_bf, (long)(rc % g.res_mod) * g.ldr + c4);
          }
#pragma unroll
          for (int rr = 0; rr < RB; rr++) {
            const int row_l = wave * RPW + rb + rr;
            const long row = m0 + row_l;
            if (row < g.M) {
              const float4 sv = *reinterpret_cast<const float4*>(stage + row_l * STAGE_LD + lane * 4);
              float v[4] = {sv.x + tab[rr].x, sv.y + tab[rr].y, sv.z + tab[rr].z, sv.w + tab[rr].w};
              if (g.gate != nullptr) {
                v[0] = gat[rr].x > 0.f ? v[0] * g.gate_scale : 0.f; v[1] = gat[rr].y > 0.f ? v[1] * g.gate_scale : 0.f;
                v[2] = gat[rr].z > 0.f ? v[2] * g.gate_scale : 0.f; v[3] = gat[rr].w > 0.f ? v[3] * g.gate_scale : 0.f;
              }
              if (g.drop_p > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; e++)
                  v[e] = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)row * g.N + c4 + e, thr) ? v[e] * inv_keep : 0.f;
              }
              v[0] += res[rr].x; v[1] += res[rr].y; v[2] += res[rr].z; v[3] += res[rr].w;
              if (g.ln_gamma != nullptr) {
                const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / BN);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 4; e++) { const float dlt = v[e] - mean; q += dlt * dlt; }
                const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / BN) + 1e-5f);
                if (g.pre_ln_out != nullptr) *reinterpret_cast<float4*>(g.pre_ln_out + row * g.ldc + c4) = make_float4(v[0], v[1], v[2], v[3]);
                v[0] = (v[0] - mean) * rstd * ga.x + be.x; v[1] = (v[1] - mean) * rstd * ga.y + be.y;
                v[2] = (v[2] - mean) * rstd * ga.z + be.z; v[3] = (v[3] - mean) * rstd * ga.w + be.w;
                if (lane == 0) {
                  if (g.ln_mean != nullptr) g.ln_mean[row] = mean;
                  if (g.ln_rstd != nullptr) g.ln_rstd[row] = rstd;
                }
              }
              store_c4(g.C, c_bf, row * g.ldc + c4, v[0], v[1], v[2], v[3]);
            }
          }
        }
        if (s + 1 < steps) {
          __syncthreads();
          wstore(wr[0], buf ^ 1);
        }
      }
    }
    __syncthreads();
    s++;
  }
}

template <int BM_, int MODE, bool EW = false, int PF = 1>
int launch_nt_as1(const hftt_gemm_nt_desc& d, hipStream_t st) {
  int lds = (BM_ * (d.K + 8) + 2 * 256 * 40) * 2;
  const int stage = BM_ * 260 * 4;
  if (MODE == 1 && stage > lds) lds = stage;
  static int attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_as1_kernel<BM_, MODE, EW, PF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { hftt_set_error("gemm_nt: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e)); return 2; }
    attr_lds = lds;
  }
  dim3 grid((unsigned)((d.M + BM_ - 1) / BM_), 1, 1);
  hipLaunchKernelGGL((gemm_nt_as1_kernel<BM_, MODE, EW, PF>), grid, dim3(512), lds, st, d);
  HFTT_CHECK_LAUNCH("gemm_nt");
  return 0;
}

template <int BN, int PREC, bool LN>
int launch_nt(const hftt_gemm_nt_desc& d, hipStream_t st) {
  using Cfg = NtCfg<BN, PREC>;
  int lds = Cfg::LOOP_BYTES;
  if ((LN || (Cfg::X3M && BN == 256)) && Cfg::STAGE_BYTES > lds) lds = Cfg::STAGE_BYTES;      // (split modes: the row-pass epilogue stages the tile)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<BN, PREC, LN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { hftt_set_error("gemm_nt: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  const int n_pad = ((d.N + 63) / 64) * 64;
  dim3 grid((unsigned)((d.M + BM - 1) / BM), (unsigned)((n_pad + BN - 1) / BN), 1);
  hipLaunchKernelGGL((gemm_nt_kernel<BN, PREC, LN>), grid, dim3(512), lds, st, d);
  HFTT_CHECK_LAUNCH("gemm_nt");
  return 0;
}

template <int PREC>
int dispatch_nt(const hftt_gemm_nt_desc& d, hipStream_t st) {
  const int n_pad = ((d.N + 63) / 64) * 64;
  if (d.ln_gamma != nullptr) {
    if (d.N == 256) return launch_nt<256, PREC, true>(d, st);
    if (d.N == 128) return launch_nt<128, PREC, true>(d, st);
    if (d.N == 64) return launch_nt<64, PREC, true>(d, st);
    hftt_set_error("gemm_nt: fused LayerNorm needs N in {64,128,256}, got %d", d.N);
    return 1;
  }
  if (n_pad % 256 == 0) return launch_nt<256, PREC, false>(d, st);
  if (n_pad % 128 == 0) return launch_nt<128, PREC, false>(d, st);
  return launch_nt<64, PREC, false>(d, st);
}

// weight-tile prefetch distance of the one-shot kernels in k steps (HFTT_NT_PF = 1 | 2 | 4 overrides for measurements)
int nt_pf() {
  static const int pf = [] { const char* e = getenv("HFTT_NT_PF"); return (e && e[0] >= '1' && e[0] <= '4') ? e[0] - '0' : 2; }();
  return pf;
}

// bf16-mode dispatch
int dispatch_nt_bf16(const hftt_gemm_nt_desc& d, hipStream_t st) {
  const bool vec_ok = (d.ldc % 4 == 0) && (((uintptr_t)d.C & 15) == 0) &&
                      (!d.residual || (d.ldr % 4 == 0 && ((uintptr_t)d.residual & 15) == 0)) &&
                      (!d.gate || (d.ldg % 4 == 0 && ((uintptr_t)d.gate & 15) == 0)) &&
                      (!d.add_table || ((uintptr_t)d.add_table & 15) == 0) &&
                      (!d.pre_ln_out || ((uintptr_t)d.pre_ln_out & 15) == 0) &&
                      (!d.ln_gamma || d.N == 256);
  if (d.N % 256 == 0 && d.K <= 768 && d.M >= 256 && vec_ok) {
    if (d.K <= 256) {       // one-shot form (measured: qkv 391 vs 518 us, o+LN 256 vs 300 us against the persistent form)
      const bool rich = d.add_table || d.gate || d.drop_p > 0.f || d.residual || d.ln_gamma;
      const int kt = d.K / 32;
      const int pf = (nt_pf() >= 4 && kt % 4 == 0) ? 4 : (nt_pf() >= 2 && kt % 2 == 0) ? 2 : 1;
      if (!rich) return pf == 4 ? launch_nt_as1<64, 0, false, 4>(d, st) : pf == 2 ? launch_nt_as1<64, 0, false, 2>(d, st) : launch_nt_as1<64, 0>(d, st);
      // dropout / bf16 ReLU gate on a bf16 C are elementwise: they ride in the direct packed-store epilogue (no staging pass)
      const bool elementwise = !d.add_table && !d.residual && !d.ln_gamma && (d.io_flags & HFTT_NT_C_BF16) && d.ldc % 2 == 0 &&
                               (!d.gate || ((d.io_flags & HFTT_NT_GATE_BF16) && d.ldg % 2 == 0));
      if (elementwise) return pf == 4 ? launch_nt_as1<64, 0, true, 4>(d, st) : pf == 2 ? launch_nt_as1<64, 0, true, 2>(d, st) : launch_nt_as1<64, 0, true>(d, st);
      if (d.N == 256) return pf == 4 ? launch_nt_as1<64, 1, false, 4>(d, st) : pf == 2 ? launch_nt_as1<64, 1, false, 2>(d, st) : launch_nt_as1<64, 1>(d, st);
      return launch_nt_as1<32, 2>(d, st);
    }
    // measured at M = 262144 (us): K=512,N=256: persistent 450 | one-shot BM=64 486;  K=768,N=256: one-shot BM=64 504 |
    // one-shot BM=32 713 | persistent double-buffered 744 | persistent 928
    if (d.K > 512 && d.N == 256) {
      const int kt = d.K / 32;
      return (nt_pf() >= 4 && kt % 4 == 0) ? launch_nt_as1<64, 1, false, 4>(d, st) : (nt_pf() >= 2 && kt % 2 == 0) ? launch_nt_as1<64, 1, false, 2>(d, st) : launch_nt_as1<64, 1>(d, st);
    }
    // K = 512, N = 256 (measured at M = 262144, us): one-shot BM=64 (1 WG/CU) 254 | one-shot BM=32 275 | persistent 288 without
    // LayerNorm; with it 354 | 340 | 369
    if (d.K <= 512 && d.N == 256 && (d.K / 32) % 2 == 0 && nt_pf() >= 2)
      return d.ln_gamma ? launch_nt_as1<32, 1, false, 2>(d, st) : launch_nt_as1<64, 1, false, 2>(d, st);
    const bool abf = d.io_flags & HFTT_NT_A_BF16;
    if (d.K <= 512) return abf ? launch_nt_as<4, false, true>(d, st) : launch_nt_as<8, false, false>(d, st);
    return abf ? launch_nt_as<6, true, true>(d, st) : launch_nt_as<12, true, false>(d, st);
  }
  if (d.io_flags & HFTT_NT_RES_BF16) { hftt_set_error("gemm_nt: a bf16 residual needs the A-stationary path (N %% 256 == 0, M >= 256, K <= 768)"); return 1; }
  return dispatch_nt<1>(d, st);      // small / ragged shapes: the k-tiled streaming kernel
}

}  // namespace

extern "C" int hftt_gemm_nt(const hftt_gemm_nt_desc* d, void* stream) {
  HFTT_REQUIRE(d != nullptr, "gemm_nt: null descriptor");
  HFTT_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_nt: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  HFTT_REQUIRE(d->K % 32 == 0, "gemm_nt: K=%d must be a multiple of 32", d->K);
  HFTT_REQUIRE((d->io_flags & ~HFTT_NT_A_HI) == 0 || d->npass == 1, "gemm_nt: bf16-stored operands need npass == 1");
  HFTT_REQUIRE(!(d->io_flags & HFTT_NT_A_HI) || d->npass == 4, "gemm_nt: HFTT_NT_A_HI goes with npass == 4");
  HFTT_REQUIRE(!(d->io_flags & HFTT_NT_C_BF16) || d->ln_gamma == nullptr, "gemm_nt: a bf16 C cannot be combined with LayerNorm");
  HFTT_REQUIRE(d->lda % ((d->io_flags & HFTT_NT_A_BF16) ? 8 : 4) == 0, "gemm_nt: lda=%ld breaks 16-byte row alignment", (long)d->lda);
  HFTT_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->W & 15) == 0, "gemm_nt: A/W must be 16-byte aligned");
  HFTT_REQUIRE(d->npass >= 1 && d->npass <= 4, "gemm_nt: npass must be 1 (bf16), 2 (split fp16), 3 (fp32) or 4 (split bf16)");
  HFTT_REQUIRE((d->npass != 2 && d->npass != 4) || d->W_lo != nullptr, "gemm_nt: the split modes need the lo weight plane (W_lo)");
  HFTT_REQUIRE(d->A != nullptr && d->W != nullptr && d->C != nullptr, "gemm_nt: null operand");
  HFTT_REQUIRE(d->add_table == nullptr || d->add_mod > 0, "gemm_nt: add_mod must be > 0");
  HFTT_REQUIRE(d->residual == nullptr || d->res_mod > 0, "gemm_nt: res_mod must be > 0");
  HFTT_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "gemm_nt: drop_p out of range");
  HFTT_REQUIRE(d->ln_gamma == nullptr || (d->ln_beta != nullptr && d->ldc == d->N), "gemm_nt: LN needs beta and ldc == N");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->npass == 3) return dispatch_nt<3>(*d, st);
  if (d->npass == 2) return dispatch_nt<2>(*d, st);
#ifdef HFTT_GRAD_HI_BUILD
  if (d->npass == 4 && (d->io_flags & HFTT_NT_A_HI)) return dispatch_nt<5>(*d, st);
#else
  HFTT_REQUIRE(!(d->io_flags & HFTT_NT_A_HI), "gemm_nt: this library was built without the gradient-rounding option (HFTT_BUILD_GRAD_HI=1 python nylon-amt_amd/build.py)");
#endif
  if (d->npass == 4) return dispatch_nt<4>(*d, st);
  return dispatch_nt_bf16(*d, st);
}
