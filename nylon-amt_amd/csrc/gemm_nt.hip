// NT GEMM with fused epilogue on MFMA (gfx950).
//   C[M,N] = epi(A[M,K] . W[N,K]^T + bias)        -- see include/hftt_hip.h (hftt_gemm_nt)
// Tile: BM=128 x BN in {64,128,256} x BK=32, 512 threads (8 waves, 2 per SIMD), double-buffered LDS,
// register-staged global->LDS.
//   npass == 1 ("bf16"):  fp32 activations are rounded to bf16 on the way into LDS, weights come as a prepared bf16
//                         plane; v_mfma_f32_32x32x16_bf16; LDS rows padded to 80 B (conflict-free ds_read_b128).
//   npass == 3 ("parity"): operands stay fp32 in LDS (rows of 33 floats: conflict-free ds_read_b32), weights come as a
//                         prepared fp32 matrix; v_mfma_f32_32x32x2_f32 = exact fp32 FMA chains (<= 1e-3 parity mode).
#include "hftt_common.h"
#include "hftt_host.h"
#include "../../include/hftt_hip.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;

template <int BN, bool F32>
struct NtCfg {
  static constexpr int WM = (BN == 64) ? 4 : 2;
  static constexpr int WN = 8 / WM;
  static constexpr int TM = BM / WM / 32;
  static constexpr int TN = BN / WN / 32;
  static constexpr int RS = F32 ? 33 : 40;                 // row stride in elements (floats / bf16)
  static constexpr int ESZ = F32 ? 4 : 2;
  static constexpr int A_ELEMS = BM * RS;
  static constexpr int W_ELEMS = BN * RS;
  static constexpr int BUF_ELEMS = A_ELEMS + W_ELEMS;
  static constexpr int LOOP_BYTES = 2 * BUF_ELEMS * ESZ;
  static constexpr int STAGE_LD = BN + 4;
  static constexpr int STAGE_BYTES = BM * STAGE_LD * 4;
  static constexpr int WCH = F32 ? (BN * 8 + 511) / 512 : (BN * 4 + 511) / 512;   // 16-byte weight chunks per thread
};

template <int BN, bool F32, bool LN>
__global__ __launch_bounds__(512) void gemm_nt_kernel(const hftt_gemm_nt_desc g) {
  using Cfg = NtCfg<BN, F32>;
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
  uint4 wreg[Cfg::WCH];
#pragma unroll
  for (int j = 0; j < Cfg::WCH; j++) wreg[j] = make_uint4(0u, 0u, 0u, 0u);
  const float* Wf = reinterpret_cast<const float*>(g.W);
  const unsigned short* Wb = reinterpret_cast<const unsigned short*>(g.W);

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = tid + 512 * j;
      const int row = i >> 3, c4 = i & 7;
      const long grow = m0 + row;
      if (grow < g.M) areg[j] = *reinterpret_cast<const float4*>(g.A + grow * g.lda + k0 + c4 * 4);
      else areg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < Cfg::WCH; j++) {
      const int i = tid + 512 * j;
      if (F32) {
        if (i < BN * 8) {
          const int row = i >> 3, ch = i & 7;
          wreg[j] = *reinterpret_cast<const uint4*>(Wf + (long)(n0 + row) * g.K + k0 + ch * 4);
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
    } else {
      unsigned short* As = sm16 + buf * Cfg::BUF_ELEMS;
      unsigned short* Ws = As + Cfg::A_ELEMS;
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
          for (int j = 0; j < TN; j++) acc[i][j] = mfma32_f32(a[i], b[j], acc[i][j]);
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
  const float inv_keep = (g.drop_p > 0.f) ? 1.0f / (1.0f - g.drop_p) : 1.0f;
  float* stage = reinterpret_cast<float*>(smem);

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
        if (ok) {
          if (g.add_table != nullptr) v += g.add_table[(long)(row % g.add_mod) * g.N + col];
          if (g.gate != nullptr) v = (g.gate[row * g.ldg + col] > 0.f) ? v * g.gate_scale : 0.f;
          if (g.drop_p > 0.f) v = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)row * g.N + col, thr) ? v * inv_keep : 0.f;
          if (g.residual != nullptr) v += g.residual[(long)(row % g.res_mod) * g.ldr + col];
        }
        if (LN) {
          stage[row_l * Cfg::STAGE_LD + col_l] = v;
        } else if (ok) {
          g.C[row * g.ldc + col] = v;
        }
      }
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

template <int BN, bool F32, bool LN>
int launch_nt(const hftt_gemm_nt_desc& d, hipStream_t st) {
  using Cfg = NtCfg<BN, F32>;
  int lds = Cfg::LOOP_BYTES;
  if (LN && Cfg::STAGE_BYTES > lds) lds = Cfg::STAGE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<BN, F32, LN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { hftt_set_error("gemm_nt: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  const int n_pad = ((d.N + 63) / 64) * 64;
  dim3 grid((unsigned)((d.M + BM - 1) / BM), (unsigned)((n_pad + BN - 1) / BN), 1);
  hipLaunchKernelGGL((gemm_nt_kernel<BN, F32, LN>), grid, dim3(512), lds, st, d);
  HFTT_CHECK_LAUNCH("gemm_nt");
  return 0;
}

template <bool F32>
int dispatch_nt(const hftt_gemm_nt_desc& d, hipStream_t st) {
  const int n_pad = ((d.N + 63) / 64) * 64;
  if (d.ln_gamma != nullptr) {
    if (d.N == 256) return launch_nt<256, F32, true>(d, st);
    if (d.N == 128) return launch_nt<128, F32, true>(d, st);
    if (d.N == 64) return launch_nt<64, F32, true>(d, st);
    hftt_set_error("gemm_nt: fused LayerNorm needs N in {64,128,256}, got %d", d.N);
    return 1;
  }
  if (n_pad % 256 == 0) return launch_nt<256, F32, false>(d, st);
  if (n_pad % 128 == 0) return launch_nt<128, F32, false>(d, st);
  return launch_nt<64, F32, false>(d, st);
}

}  // namespace

extern "C" int hftt_gemm_nt(const hftt_gemm_nt_desc* d, void* stream) {
  HFTT_REQUIRE(d != nullptr, "gemm_nt: null descriptor");
  HFTT_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_nt: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  HFTT_REQUIRE(d->K % 32 == 0, "gemm_nt: K=%d must be a multiple of 32", d->K);
  HFTT_REQUIRE(d->lda % 4 == 0, "gemm_nt: lda=%ld must be a multiple of 4", (long)d->lda);
  HFTT_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->W & 15) == 0, "gemm_nt: A/W must be 16-byte aligned");
  HFTT_REQUIRE(d->npass == 1 || d->npass == 3, "gemm_nt: npass must be 1 (bf16) or 3 (fp32 parity)");
  HFTT_REQUIRE(d->A != nullptr && d->W != nullptr && d->C != nullptr, "gemm_nt: null operand");
  HFTT_REQUIRE(d->add_table == nullptr || d->add_mod > 0, "gemm_nt: add_mod must be > 0");
  HFTT_REQUIRE(d->residual == nullptr || d->res_mod > 0, "gemm_nt: res_mod must be > 0");
  HFTT_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "gemm_nt: drop_p out of range");
  HFTT_REQUIRE(d->ln_gamma == nullptr || (d->ln_beta != nullptr && d->ldc == d->N), "gemm_nt: LN needs beta and ldc == N");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->npass == 3) return dispatch_nt<true>(*d, st);
  return dispatch_nt<false>(*d, st);
}
