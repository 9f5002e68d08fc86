// Strip kernels of the bf16 mode for SMALL widths (round 5): every nn.Linear with K, N <= 192 and the fused position-wise feed-forward block
// d = 64 / p = 128 of the reference's default model (training/m_training.py:56-61) -- BASELINE config 2, "Tiny hFT bf16".  Until round 5 the
// bf16 mode at this size ran the round-1 block GEMMs (51 % of its step) and was slower than the x3 mode, whose small-width family
// (x3s_strip.h) this file mirrors on a bf16 activation stream:
//   * the WHOLE weight matrix of a launch lives in LDS (<= 24 KB as bf16 fragments; 32 KB for the fused block's two): no ring, no per-slot
//     barrier -- a workgroup copies the pack once and walks 128-token blocks, each wave on its own 32-token strip (MFMA B operand, 4 registers
//     per 16-feature chunk); the weights are the A operand, so accumulator register g of a lane is feature 16 h + g of the lane's token and
//     bias / ReLU / dropout / residual / LayerNorm run per lane;
//   * every tensor between kernels is bf16; results leave as whole 128-byte lines through a wave-private LDS patch (strip_pipe.h: patch_put /
//     patch_flush -- a PAIR of 32-column tiles at a time; every N of this family is a multiple of 64);
//   * one MFMA pass (bf16 operands, fp32 accumulate).
// Pack: hftt_x3_strip_pack order 2 with bf16 halves (element type 4) -- the pair of (k chunk c, output tile t) at pair index
// slot_offset + c * NT + t, 2 KB per pair: this family reads the hi fragment (the bf16 rounding of the weight) and skips the lo one, so one
// pack serves both precision modes of the small model.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "hftt_common.h"
#include "hftt_host.h"
#include "strip_internal.h"
#include "../../include/hftt_hip.h"

namespace {

#include "strip_pipe.h"

// element offset, inside the lane's row view (row + 16 * h), of chunk ch = 2 * pt + u: features 32 * pt + 8 * u + [0, 8) of the lane half
__device__ __forceinline__ int bs_chunk_off(int ch) { return (ch >> 1) * 32 + (ch & 1) * 8; }

// hi fragments of `pairs` (hi, lo) pairs -> LDS, 1 KB each (16 bytes per thread and step)
__device__ __forceinline__ void bs_copy_weights(unsigned char* lds, const unsigned short* w, int pairs, int tid) {
  const unsigned char* src = reinterpret_cast<const unsigned char*>(w);
  for (int i = tid; i < pairs * 64; i += 256) {
    const int pr = i >> 6, c = i & 63;
    *reinterpret_cast<uint4*>(lds + pr * 1024 + c * 16) = *reinterpret_cast<const uint4*>(src + (long)pr * 2048 + c * 16);
  }
}
__device__ __forceinline__ bf16x8 bs_frag(const unsigned char* wl, int idx) { return *reinterpret_cast<const bf16x8*>(wl + idx * 1024); }

__device__ __forceinline__ void bs_load16h(const unsigned short* p, float* v) {      // 16 bf16 -> fp32
  const u4v a = *reinterpret_cast<const u4v*>(p), b = *reinterpret_cast<const u4v*>(p + 8);
  unpack8(a, v); unpack8(b, v + 8);
}

// LayerNorm over the 64 features of the lane's token (32 in acc, partner lane ^ 32 the rest); pre-LayerNorm rows and the output leave through the patch
__device__ __forceinline__ void bs_ln_rows(f32x16 (&acc)[2], const float* gamma_lds, const float* beta_lds, int j, int h, int lane, float* mean_out, float* rstd_out,
                                           long tok, bool ok, unsigned char* patch, const unsigned short* pre_wave, const unsigned short* y_wave, int ld) {
  float s = 0.f;
#pragma unroll
  for (int ot = 0; ot < 2; ot++)
#pragma unroll
    for (int q = 0; q < 16; q++) s += acc[ot][q];
  const float mean = xor32_sum(s) * (1.0f / 64.0f);
  float qs = 0.f;
#pragma unroll
  for (int ot = 0; ot < 2; ot++)
#pragma unroll
    for (int q = 0; q < 16; q++) { const float dlt = acc[ot][q] - mean; qs += dlt * dlt; }
  const float rstd = 1.0f / sqrtf(xor32_sum(qs) * (1.0f / 64.0f) + 1e-5f);
  if (ok && h == 0) {
    if (mean_out != nullptr) mean_out[tok] = mean;
    if (rstd_out != nullptr) rstd_out[tok] = rstd;
  }
  if (pre_wave != nullptr && ok) {                    // (wave-uniform)
#pragma unroll
    for (int ot = 0; ot < 2; ot++) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = acc[ot][q];
      patch_put(patch, j, h, ot, pack8u(v), pack8u(v + 8));
    }
    patch_flush(patch, lane, pre_wave, ld, 0);
  }
#pragma unroll
  for (int ot = 0; ot < 2; ot++) {
    float v[16], ga[16], be[16];
    lds16f(gamma_lds + ot * 32 + 16 * h, ga);
    lds16f(beta_lds + ot * 32 + 16 * h, be);
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = (acc[ot][q] - mean) * rstd * ga[q] + be[q];
    if (ok) patch_put(patch, j, h, ot, pack8u(v), pack8u(v + 8));
  }
  if (ok) patch_flush(patch, lane, y_wave, ld, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// C = epi(x . Wl^T + bias): K = 32 * K32, N = 32 * NT (NT even);  LN: N == 64 with dropout / residual / LayerNorm
// ---------------------------------------------------------------------------------------------------------------------
template <int K32, int NT, bool LN>
struct BsCfg {
  static constexpr int KC = 2 * K32;
  static constexpr int WBYTES = KC * NT * 1024;
  static constexpr int PRM = NT * 32 + (LN ? 128 : 0);                  // bias | gamma | beta (floats)
  static constexpr int LDS = WBYTES + 4 * PRM + 4 * PATCH_BYTES;
};

template <int K32, int NT, bool LN, bool HR>
__global__ __launch_bounds__(256, 2) void bs_linear_kernel(const hftt_strip_desc g) {
  using Cfg = BsCfg<K32, NT, LN>;
  constexpr int KC = Cfg::KC;
  static_assert(NT % 2 == 0 && (!LN || NT == 2), "tile pairs; LayerNorm form: N == 64");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + Cfg::WBYTES);
  unsigned char* patch = smem + Cfg::WBYTES + 4 * Cfg::PRM + wave * PATCH_BYTES;
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(g.x);
  const unsigned short* cb = reinterpret_cast<const unsigned short*>(g.C);
  const unsigned short* preb = reinterpret_cast<const unsigned short*>(g.pre_ln_out);
  const unsigned short* rb = reinterpret_cast<const unsigned short*>(g.residual);
  const bool relu = g.flags & HFTT_SL_RELU;

  bs_copy_weights(smem, g.w, KC * NT, tid);
  for (int i = tid; i < NT * 32; i += 256) prm[i] = g.bias != nullptr ? g.bias[i] : 0.f;
  if (LN && tid < 64) { prm[NT * 32 + tid] = g.ln_gamma[tid]; prm[NT * 32 + 64 + tid] = g.ln_beta[tid]; }
  __syncthreads();

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* wl = smem + lane * 16;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;
    asm volatile("" : "+v"(hb));                       // (per-tile column arithmetic stays inside the iteration)
    const long tok = blk * 128 + wave * 32 + j;
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;            // M % 32 == 0 (host check)
    const long tokc = tok < g.M ? tok : (long)g.M - 1;
    u4v xr[KC];
    {
      const unsigned short* p0 = xb + tokc * g.ldx + 16 * hb;
#pragma unroll
      for (int c = 0; c < KC; c++) xr[c] = *reinterpret_cast<const u4v*>(p0 + bs_chunk_off(c));
    }
    const long rrow = g.res_mod > 0 ? (long)((unsigned)tokc % (unsigned)g.res_mod) : tokc;
    const unsigned short* rrow_p = rb + (HR ? rrow * g.ldr + 16 * hb : 0);
    const unsigned short* cwave = cb + (blk * 128 + wave * 32) * g.ldc;
    const uint64_t rowq = ((uint64_t)tok * (uint64_t)g.N) >> 2;
    f32x16 lacc[LN ? 2 : 1];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      f32x16 acc;
      {
        float b[16];
        lds16f(prm + t * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = b[q];
      }
      float r[16];
      if (HR) bs_load16h(rrow_p + t * 32, r);
#pragma unroll
      for (int c = 0; c < KC; c++) acc = mfma32(bs_frag(wl, c * NT + t), as_frag(xr[c]), acc);
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        float a = acc[q];
        if (!LN && relu) a = fmaxf(a, 0.f);
        v[q] = a * g.out_scale;
      }
      if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.drop_site, rowq + ((t * 32 + 16 * hb) >> 2), thr, inv_keep);
      if (HR) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] += r[q];
      }
      if constexpr (LN) {
#pragma unroll
        for (int q = 0; q < 16; q++) lacc[t][q] = v[q];
      } else if (wave_ok) {
        patch_put(patch, j, hb, t & 1, pack8u(v), pack8u(v + 8));
        if (t & 1) patch_flush(patch, lane, cwave, g.ldc, t >> 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LN) {
      bs_ln_rows(lacc, prm + NT * 32, prm + NT * 32 + 64, j, hb, lane, g.ln_mean, g.ln_rstd, tok, wave_ok, patch,
                 preb != nullptr ? preb + (blk * 128 + wave * 32) * g.ldc : nullptr, cwave, g.ldc);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fused two-GEMM block, d = 64, p = 128: mode 0 = FFN forward + residual + LayerNorm, mode 1 = dX half of its backward.
// Weights: first matrix fragments (k chunk c of 4, hidden tile t of 4) at c * 4 + t, second matrix (k chunk c of 8, output tile ot of 2) at
// 16 + c * 2 + ot: 32 KB; one patch per wave: three workgroups per CU by LDS, two by the launch bound.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BSM_WBYTES = 32 * 1024;
constexpr int BSM_PRM = 128 + 64 + 128;                                  // b1 | b2 | gamma | beta
constexpr int BSM_LDS = BSM_WBYTES + 4 * BSM_PRM + 4 * PATCH_BYTES;

template <int MODE>
__global__ __launch_bounds__(256, 2) void bs_mlp_kernel(const hftt_ffn_desc g) {
  constexpr int PT = 4, p = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + BSM_WBYTES);
  unsigned char* patch = smem + BSM_WBYTES + 4 * BSM_PRM + wave * PATCH_BYTES;
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(g.x);
  const unsigned short* yb = reinterpret_cast<const unsigned short*>(g.y);
  const unsigned short* preb = reinterpret_cast<const unsigned short*>(g.pre_ln_out);
  const unsigned short* rb = reinterpret_cast<const unsigned short*>(g.residual);
  const unsigned short* hob = reinterpret_cast<const unsigned short*>(g.h_out);
  const unsigned short* gtb = reinterpret_cast<const unsigned short*>(g.gate);
  const bool has_res = (MODE == 1) && g.residual != nullptr;

  bs_copy_weights(smem, g.w, 32, tid);
  if (tid < 128) prm[tid] = (MODE == 0 && g.b1 != nullptr) ? g.b1[tid] : 0.f;
  if (tid < 64) {
    prm[128 + tid] = (MODE == 0 && g.b2 != nullptr) ? g.b2[tid] : 0.f;
    if (MODE == 0) { prm[192 + tid] = g.ln_gamma[tid]; prm[256 + tid] = g.ln_beta[tid]; }
  }
  __syncthreads();

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* wl = smem + lane * 16;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;
    asm volatile("" : "+v"(hb));
    const long tok = blk * 128 + wave * 32 + j;
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;
    const long tokc = tok < g.M ? tok : (long)g.M - 1;
    u4v xr[4];
    {
      const unsigned short* p0 = xb + tokc * g.ldx + 16 * hb;
#pragma unroll
      for (int c = 0; c < 4; c++) xr[c] = *reinterpret_cast<const u4v*>(p0 + bs_chunk_off(c));
    }
    const uint64_t rowq_h = ((uint64_t)tok * (uint64_t)p) >> 2;
    const unsigned short* hwave = hob + (blk * 128 + wave * 32) * g.ldh;
    f32x16 yacc[2];
#pragma unroll
    for (int ot = 0; ot < 2; ot++) {
      float b[16];
      lds16f(prm + 128 + ot * 32 + 16 * hb, b);
#pragma unroll
      for (int q = 0; q < 16; q++) yacc[ot][q] = b[q];
    }
#pragma unroll
    for (int t = 0; t < PT; t++) {
      // ---- first GEMM, hidden tile t ----
      f32x16 hacc;
      {
        float b[16];
        lds16f(prm + t * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) hacc[q] = b[q];
      }
      float gcur[16];
      if (MODE == 1) bs_load16h(gtb + tokc * g.ldg + t * 32 + 16 * hb, gcur);
#pragma unroll
      for (int c = 0; c < 4; c++) hacc = mfma32(bs_frag(wl, c * 4 + t), as_frag(xr[c]), hacc);
      // ---- middle epilogue: the lane's 16 hidden features of tile t become the B operand of the second GEMM ----
      float v[16];
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = fmaxf(hacc[q], 0.f);
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_h, rowq_h + ((t * 32 + 16 * hb) >> 2), thr, inv_keep);
      } else {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = gcur[q] > 0.f ? hacc[q] * g.gate_scale : 0.f;
      }
      u4v hf[2];
      hf[0] = pack8u(v); hf[1] = pack8u(v + 8);
      if (hob != nullptr && wave_ok) {                                       // (wave-uniform)
        patch_put(patch, j, hb, t & 1, hf[0], hf[1]);
        if (t & 1) patch_flush(patch, lane, hwave, g.ldh, t >> 1);
      }
      // ---- second GEMM, K-slice t (chunks 2t, 2t + 1) ----
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int ot = 0; ot < 2; ot++) yacc[ot] = mfma32(bs_frag(wl, 16 + (2 * t + u) * 2 + ot), as_frag(hf[u]), yacc[ot]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---------------- final epilogue of the block ----------------
    const uint64_t rowq = ((uint64_t)tok * 64ull) >> 2;
    const unsigned short* ywave = yb + (blk * 128 + wave * 32) * g.ldy;
    const unsigned short* rrow_p = rb + (has_res ? tokc * g.ldr + 16 * hb : 0);
#pragma unroll
    for (int ot = 0; ot < 2; ot++) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = yacc[ot][q];
      if (MODE == 0 && g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_o, rowq + ((ot * 32 + 16 * hb) >> 2), thr, inv_keep);
      if (MODE == 0) {                                  // residual = the block input, still in the strip registers
        float r[16];
        unpack8(xr[2 * ot], r); unpack8(xr[2 * ot + 1], r + 8);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] += r[q];
      } else if (has_res) {
        float r[16];
        bs_load16h(rrow_p + ot * 32, r);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] += r[q];
      }
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) yacc[ot][q] = v[q];
      } else if (wave_ok) {
        patch_put(patch, j, hb, ot, pack8u(v), pack8u(v + 8));
      }
    }
    if (MODE == 0) {
      bs_ln_rows(yacc, prm + 192, prm + 256, j, hb, lane, g.ln_mean, g.ln_rstd, tok, wave_ok, patch,
                 preb != nullptr ? preb + (blk * 128 + wave * 32) * g.ldy : nullptr, ywave, g.ldy);
    } else if (wave_ok) {
      patch_flush(patch, lane, ywave, g.ldy, 0);
    }
  }
}

int bs_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n = prop.multiProcessorCount;
  }
  return n;
}
template <typename K>
int bs_set_lds(K kernel, int lds, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { hftt_set_error("%s: hipFuncSetAttribute(%d B LDS) failed: %s", what, lds, hipGetErrorString(e)); return 2; }
  return 0;
}
template <int K32, int NT, bool LN, bool HR>
int launch_bs(const hftt_strip_desc& d, hipStream_t st) {
  using Cfg = BsCfg<K32, NT, LN>;
  static bool attr = false;
  if (!attr) { if (int rc = bs_set_lds(bs_linear_kernel<K32, NT, LN, HR>, Cfg::LDS, "bs_strip_linear")) return rc; attr = true; }
  const int cus = bs_cus();
  if (cus <= 0) { hftt_set_error("bs_strip_linear: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  const long cap = 2L * cus;
  hipLaunchKernelGGL((bs_linear_kernel<K32, NT, LN, HR>), dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(256), Cfg::LDS, st, d);
  HFTT_CHECK_LAUNCH("bs_strip_linear");
  return 0;
}
template <int MODE>
int launch_bsm(const hftt_ffn_desc& d, hipStream_t st) {
  static bool attr = false;
  if (!attr) { if (int rc = bs_set_lds(bs_mlp_kernel<MODE>, BSM_LDS, "bs_strip_mlp")) return rc; attr = true; }
  const int cus = bs_cus();
  if (cus <= 0) { hftt_set_error("bs_strip_mlp: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((bs_mlp_kernel<MODE>), dim3((unsigned)(nblk < 2L * cus ? nblk : 2L * cus)), dim3(256), BSM_LDS, st, d);
  HFTT_CHECK_LAUNCH("bs_strip_mlp");
  return 0;
}

}  // namespace

// -1: not a shape / storage of this family (the caller goes on to its own checks); otherwise the launch status.
// The shapes of the d = 64 model (K x N): forward 64x192 (q, k, v), 64x128 (cross k, v), 64x64 (cross q; fc_o + LayerNorm), backward 64x64,
// 192x64 / 128x64 (dX of the fused projections, + residual)
int hftt_bs_strip_linear_try(const hftt_strip_desc& d, hipStream_t st) {
  const uint32_t bf = HFTT_SL_X_BF16 | HFTT_SL_C_BF16;
  if ((d.flags & bf) != bf || d.N > 192 || d.K > 192 || d.N % 64 != 0 || d.K % 64 != 0) return -1;
  if (d.M <= 0 || d.M % 32 != 0 || d.x == nullptr || d.w == nullptr || d.C == nullptr || d.gate != nullptr) {
    hftt_set_error("bs_strip_linear: M=%d must be a positive multiple of 32, x / w / C non-null, no gate", d.M);
    return 1;
  }
  if (d.residual != nullptr && !(d.flags & HFTT_SL_RES_BF16)) { hftt_set_error("bs_strip_linear: the residual of the bf16 small-width family is bf16"); return 1; }
  if ((((uintptr_t)d.x | (uintptr_t)d.C | (uintptr_t)d.w | (uintptr_t)d.residual | (uintptr_t)d.pre_ln_out) & 15) != 0 || d.ldx % 8 != 0 || d.ldc % 8 != 0 ||
      (d.residual != nullptr && d.ldr % 8 != 0)) {
    hftt_set_error("bs_strip_linear: operands must be 16-byte aligned with row strides that are multiples of 8 elements");
    return 1;
  }
  if (d.drop_p < 0.f || d.drop_p >= 1.f || d.res_mod < 0 || (long)d.M * d.N >= (1L << 33)) { hftt_set_error("bs_strip_linear: drop_p / res_mod / M*N out of range"); return 1; }
  const int k32 = d.K / 32, nt = d.N / 32;
  const bool hr = d.residual != nullptr;
  if (d.ln_gamma != nullptr) {
    if (d.N != 64 || d.ln_beta == nullptr || d.ldc != 64 || (d.flags & HFTT_SL_RELU)) { hftt_set_error("bs_strip_linear: LayerNorm needs N == ldc == 64, beta, no ReLU"); return 1; }
    if (k32 == 2) return hr ? launch_bs<2, 2, true, true>(d, st) : launch_bs<2, 2, true, false>(d, st);
  } else if (k32 == 2) {
    if (nt == 2) return hr ? launch_bs<2, 2, false, true>(d, st) : launch_bs<2, 2, false, false>(d, st);
    if (nt == 4 && !hr) return launch_bs<2, 4, false, false>(d, st);
    if (nt == 6 && !hr) return launch_bs<2, 6, false, false>(d, st);
  } else if (nt == 2) {
    if (k32 == 4) return hr ? launch_bs<4, 2, false, true>(d, st) : launch_bs<4, 2, false, false>(d, st);
    if (k32 == 6) return hr ? launch_bs<6, 2, false, true>(d, st) : launch_bs<6, 2, false, false>(d, st);
  }
  hftt_set_error("bs_strip_linear: shape N=%d K=%d%s is not covered (K x N in {64x64, 64x128, 64x192, 128x64, 192x64}; LayerNorm: 64x64)", d.N, d.K,
                 d.ln_gamma != nullptr ? " with LayerNorm" : "");
  return 1;
}

int hftt_bs_strip_mlp_try(const hftt_ffn_desc& d, hipStream_t st) {
  if (d.d != 64 || d.p != 128) return -1;
  // the family computes on the bf16 stream only: the same storage requirement as check_ffn (strip_gemm.hip) -- a d = 64 descriptor without
  // the all-bf16 flags is an error here, not a reinterpretation of fp32 buffers
  const uint32_t all = HFTT_SL_X_BF16 | HFTT_SL_C_BF16 | HFTT_SL_RES_BF16;
  if ((d.flags & all) != all) { hftt_set_error("bs_strip_mlp: the fused block is all-bf16 (flags X_BF16 | C_BF16 | RES_BF16), got flags 0x%x", d.flags); return 1; }
  if ((long)d.M * d.p >= (1L << 33)) { hftt_set_error("bs_strip_mlp: M*p too large for the 32-bit dropout pair index"); return 1; }
  if (d.M <= 0 || d.M % 32 != 0 || d.x == nullptr || d.w == nullptr || d.y == nullptr) { hftt_set_error("bs_strip_mlp: M=%d must be a positive multiple of 32, x / w / y non-null", d.M); return 1; }
  if ((((uintptr_t)d.x | (uintptr_t)d.y | (uintptr_t)d.w | (uintptr_t)d.residual | (uintptr_t)d.pre_ln_out | (uintptr_t)d.h_out | (uintptr_t)d.gate) & 15) != 0 ||
      d.ldx % 8 != 0 || d.ldy % 8 != 0 || (d.h_out != nullptr && d.ldh % 8 != 0) || (d.gate != nullptr && d.ldg % 8 != 0) || (d.residual != nullptr && d.ldr % 8 != 0)) {
    hftt_set_error("bs_strip_mlp: operands must be 16-byte aligned with row strides that are multiples of 8 elements");
    return 1;
  }
  if (d.drop_p < 0.f || d.drop_p >= 1.f) { hftt_set_error("bs_strip_mlp: drop_p out of range"); return 1; }
  if (d.mode == 0) {
    if (d.ln_gamma == nullptr || d.ln_beta == nullptr || d.ldy != 64 || d.residual != nullptr) { hftt_set_error("bs_strip_mlp: the forward needs gamma, beta, ldy == 64 and takes its residual from x"); return 1; }
    return launch_bsm<0>(d, st);
  }
  if (d.gate == nullptr || d.h_out == nullptr) { hftt_set_error("bs_strip_mlp: the backward needs the stored hidden (gate) and the dh output"); return 1; }
  return launch_bsm<1>(d, st);
}
