// Strip kernels of the split-operand ("x3") mode for SMALL widths (the reference's default model, training/m_training.py:56-61: d = 64,
// ff = 128): every nn.Linear with K, N <= 192 and the fused position-wise feed-forward block d = 64 / p = 128.  Included inside the
// anonymous namespace of x3_strip.hip (it uses that file's strip chunks, row-segment stores and dropout helpers).
//
// What is different from the d = 256 family: the WHOLE weight matrix of a launch fits LDS (<= 48 KB as fp16 / bf16 hi + lo fragment pairs;
// 64 KB for the fused block), so there is no ring, no per-slot barrier and no weight traffic after the first microsecond -- a workgroup
// copies the packed stream once and then walks 128-token blocks, each wave on its own 32-token strip.  These launches are HBM-bound by a
// wide margin (arithmetic intensity <= 128 F/B at d = 64), so the design goal is bytes in flight: small register footprints (strip
// 32 .. 96 registers, ONE accumulator tile) for 2-3 workgroups per CU, whole-line row-segment stores, nothing else.
//
// Pack: hftt_x3_strip_pack order 2 ("compact"): fragment pair (k chunk c, output tile t) at pair index slot_offset + c * NT + t
// (slot_stride = NT = N / 32), 2 KB per pair: 1 KB hi fragment, 1 KB lo fragment, lane layout as in the other orders.
#pragma once

template <int NTL>
__device__ __forceinline__ void x3s_ln_rows(f32x16 (&acc)[NTL], const float* gamma_lds, const float* beta_lds, int h, float* mean_out, float* rstd_out,
                                            long tok, bool ok, float* stage, int j, int lane, float* pre_wave, float* y_wave, long ld,
                                            unsigned short* pre16_wave) {
  constexpr float RN = 1.0f / (32.0f * NTL);
  float s = 0.f;
#pragma unroll
  for (int ot = 0; ot < NTL; ot++)
#pragma unroll
    for (int q = 0; q < 16; q++) s += acc[ot][q];
  const float mean = xor32_sum(s) * RN;
  float qs = 0.f;
#pragma unroll
  for (int ot = 0; ot < NTL; ot++)
#pragma unroll
    for (int q = 0; q < 16; q++) { const float dlt = acc[ot][q] - mean; qs += dlt * dlt; }
  const float rstd = 1.0f / sqrtf(xor32_sum(qs) * RN + 1e-5f);
  if (ok && h == 0) {
    if (mean_out != nullptr) mean_out[tok] = mean;
    if (rstd_out != nullptr) rstd_out[tok] = rstd;
  }
#pragma unroll
  for (int ot = 0; ot < NTL; ot++) {
    float v[16], ga[16], be[16];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = acc[ot][q];
    if (pre_wave != nullptr) tile_store_rows(stage, v, j, h, lane, pre_wave + ot * 32, ld, ok);
    if (pre16_wave != nullptr) tile_store_rows_bf16(stage, v, j, h, lane, pre16_wave + ot * 32, ld, ok);
    lds16f(gamma_lds + ot * 32 + 16 * h, ga);
    lds16f(beta_lds + ot * 32 + 16 * h, be);
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = (v[q] - mean) * rstd * ga[q] + be[q];
    tile_store_rows(stage, v, j, h, lane, y_wave + ot * 32, ld, ok);
  }
}

// packed stream -> LDS, 16 bytes per thread and step (every wave-instruction a contiguous kilobyte)
__device__ __forceinline__ void x3s_copy_weights(unsigned char* lds, const unsigned short* w, int bytes, int tid) {
  const uint4* src = reinterpret_cast<const uint4*>(w);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (int i = tid; i < bytes / 16; i += 256) dst[i] = src[i];
}

// acc += W(tile, all KC chunks) . strip: three passes per chunk, small terms first (the order of x3_slot_tiles)
template <int E, int KC, int NT>
__device__ __forceinline__ void x3s_tile(const unsigned char* wl, int t, const XChunk (&x)[KC], f32x16& acc) {
#pragma unroll
  for (int c = 0; c < KC; c++) {
    const unsigned char* p = wl + (c * NT + t) * 2048;
    const bf16x8 fh = *reinterpret_cast<const bf16x8*>(p), fl = *reinterpret_cast<const bf16x8*>(p + 1024);
    const bf16x8 xh = __builtin_bit_cast(bf16x8, x[c].a), xl = __builtin_bit_cast(bf16x8, x[c].b);
    acc = X3<E>::mma(fl, xh, acc);
    acc = X3<E>::mma(fh, xl, acc);
    acc = X3<E>::mma(fh, xh, acc);
  }
}

constexpr int x3s_wgs(int lds_bytes) { return lds_bytes <= 52 * 1024 ? 3 : (lds_bytes <= 78 * 1024 ? 2 : 1); }

// ---------------------------------------------------------------------------------------------------------------------
// C = epi(x . Wl^T + bias): K = 32 * K32, N = 32 * NT;  LN: N == 64 with dropout / residual / LayerNorm
// ---------------------------------------------------------------------------------------------------------------------
template <int K32, int NT, bool LN>
struct XsCfg {
  static constexpr int KC = 2 * K32;
  static constexpr int WBYTES = KC * NT * 2048;
  static constexpr int PRM = NT * 32 + (LN ? 128 : 0);                 // bias | gamma | beta (floats)
  static constexpr int LDS = WBYTES + 4 * PRM + 4 * STG_BYTES_PER_WAVE;
};

template <int E, int K32, int NT, bool LN, bool HR>
__global__ __launch_bounds__(256, x3s_wgs(XsCfg<K32, NT, LN>::LDS)) void x3s_linear_kernel(const hftt_strip_desc g) {
  using Cfg = XsCfg<K32, NT, LN>;
  constexpr int KC = Cfg::KC;
  static_assert(!LN || NT == 2, "LayerNorm form: N == 64");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + Cfg::WBYTES);
  float* stage = reinterpret_cast<float*>(smem + Cfg::WBYTES + 4 * Cfg::PRM + wave * STG_BYTES_PER_WAVE);
  const float* xb = reinterpret_cast<const float*>(g.x);
  float* cb = reinterpret_cast<float*>(g.C);
  float* preb = reinterpret_cast<float*>(g.pre_ln_out);
  const float* rb = reinterpret_cast<const float*>(g.residual);
  const bool relu = g.flags & HFTT_SL_RELU;

  x3s_copy_weights(smem, g.w, Cfg::WBYTES, tid);
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
    XChunk xr[KC];
    {
      const float* p0 = xb + tokc * g.ldx + 16 * hb;
#pragma unroll
      for (int c = 0; c < KC; c++) chunk_load(xr[c], p0 + chunk_off(c));
#pragma unroll
      for (int c = 0; c < KC; c++) chunk_convert<E>(xr[c]);
    }
    const long rrow = g.res_mod > 0 ? (long)((unsigned)tokc % (unsigned)g.res_mod) : tokc;
    const float* rrow_p = rb + (HR ? rrow * g.ldr + 16 * hb : 0);
    float* cwave = cb + (blk * 128 + wave * 32) * g.ldc;
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
      if (HR) load16f(rrow_p + t * 32, r);
      x3s_tile<E, KC, NT>(wl, t, xr, acc);
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
      } else {
        tile_store_rows(stage, v, j, hb, lane, cwave + t * 32, g.ldc, wave_ok);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LN) {
      const bool p16 = g.flags & HFTT_SL_PRE_BF16;
      x3s_ln_rows<2>(lacc, prm + NT * 32, prm + NT * 32 + 64, hb, g.ln_mean, g.ln_rstd, tok, wave_ok, stage, j, lane,
                     (preb != nullptr && !p16) ? preb + (blk * 128 + wave * 32) * g.ldc : nullptr, cwave, g.ldc,
                     (preb != nullptr && p16) ? reinterpret_cast<unsigned short*>(preb) + (blk * 128 + wave * 32) * g.ldc : nullptr);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fused two-GEMM block, d = 64, p = 128: mode 0 = FFN forward + residual + LayerNorm (fp16 halves), mode 1 = dX half of its backward
// (bf16 halves).  Weights: first matrix pairs (k chunk c of 4, hidden tile t of 4) at pair c * 4 + t, second matrix pairs (k chunk c of 8,
// output tile ot of 2) at pair 16 + c * 2 + ot: 64 KB, two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int XSM_WBYTES = 32 * 2048;
constexpr int XSM_PRM = 128 + 64 + 128;                                  // b1 | b2 | gamma | beta
constexpr int XSM_LDS = XSM_WBYTES + 4 * XSM_PRM + 4 * STG_BYTES_PER_WAVE;

template <int MODE, bool HH>
__global__ __launch_bounds__(256, 2) void x3s_mlp_kernel(const hftt_ffn_desc g) {
  constexpr int E = (MODE == 0) ? X3_F16 : X3_BF16;
  constexpr int PT = 4, p = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + XSM_WBYTES);
  float* stage = reinterpret_cast<float*>(smem + XSM_WBYTES + 4 * XSM_PRM + wave * STG_BYTES_PER_WAVE);
  const float* xb = reinterpret_cast<const float*>(g.x);
  float* yb = reinterpret_cast<float*>(g.y);
  float* preb = reinterpret_cast<float*>(g.pre_ln_out);
  const float* rb = reinterpret_cast<const float*>(g.residual);
  float* hob = reinterpret_cast<float*>(g.h_out);
  const float* gtb = reinterpret_cast<const float*>(g.gate);
  unsigned short* hob16 = reinterpret_cast<unsigned short*>(g.h_out);
  const unsigned short* gtb16 = reinterpret_cast<const unsigned short*>(g.gate);
  const bool has_res = (MODE == 1) && g.residual != nullptr;

  x3s_copy_weights(smem, g.w, XSM_WBYTES, tid);
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
    XChunk xr[4];
    {
      const float* p0 = xb + tokc * g.ldx + 16 * hb;
#pragma unroll
      for (int c = 0; c < 4; c++) chunk_load(xr[c], p0 + chunk_off(c));
#pragma unroll
      for (int c = 0; c < 4; c++) chunk_convert<E>(xr[c]);
    }
    const uint64_t rowq_h = ((uint64_t)tok * (uint64_t)p) >> 2;
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
      if (MODE == 1) { if (HH) load16h(gtb16 + tokc * g.ldg + t * 32 + 16 * hb, gcur); else load16f(gtb + tokc * g.ldg + t * 32 + 16 * hb, gcur); }
      x3s_tile<E, 4, 4>(wl, t, xr, hacc);
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
      XChunk hf[2];
      {
        bf16x8 hi, lo;
        x3_split8<E>(v, hi, lo);
        hf[0].a = __builtin_bit_cast(u4v, hi); hf[0].b = __builtin_bit_cast(u4v, lo);
        x3_split8<E>(v + 8, hi, lo);
        hf[1].a = __builtin_bit_cast(u4v, hi); hf[1].b = __builtin_bit_cast(u4v, lo);
      }
      if (hob != nullptr) {                                                  // (wave-uniform)
        if (HH) tile_store_rows_bf16(stage, v, j, hb, lane, hob16 + (blk * 128 + wave * 32) * g.ldh + t * 32, g.ldh, wave_ok);
        else tile_store_rows(stage, v, j, hb, lane, hob + (blk * 128 + wave * 32) * g.ldh + t * 32, g.ldh, wave_ok);
      }
      // ---- second GEMM, K-slice t (chunks 2t, 2t + 1) ----
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int ot = 0; ot < 2; ot++) {
          const unsigned char* pw = wl + (16 + (2 * t + u) * 2 + ot) * 2048;
          const bf16x8 fh = *reinterpret_cast<const bf16x8*>(pw), fl = *reinterpret_cast<const bf16x8*>(pw + 1024);
          const bf16x8 xh = __builtin_bit_cast(bf16x8, hf[u].a), xl = __builtin_bit_cast(bf16x8, hf[u].b);
          yacc[ot] = X3<E>::mma(fl, xh, yacc[ot]);
          yacc[ot] = X3<E>::mma(fh, xl, yacc[ot]);
          yacc[ot] = X3<E>::mma(fh, xh, yacc[ot]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---------------- final epilogue of the block ----------------
    const uint64_t rowq = ((uint64_t)tok * 64ull) >> 2;
    float* ywave = yb + (blk * 128 + wave * 32) * g.ldy;
    const float* rrow_p = rb + (has_res ? tokc * g.ldr + 16 * hb : 0);
#pragma unroll
    for (int ot = 0; ot < 2; ot++) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = yacc[ot][q];
      if (MODE == 0 && g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_o, rowq + ((ot * 32 + 16 * hb) >> 2), thr, inv_keep);
      if (MODE == 0) {                                  // residual = the block input, still in the strip registers (hi + lo)
        float r[16];
        chunk_values<E>(xr[2 * ot], r); chunk_values<E>(xr[2 * ot + 1], r + 8);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] += r[q];
      } else if (has_res) {
        float r[16];
        load16f(rrow_p + ot * 32, r);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] += r[q];
      }
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) yacc[ot][q] = v[q];
      } else {
        tile_store_rows(stage, v, j, hb, lane, ywave + ot * 32, g.ldy, wave_ok);
      }
    }
    if (MODE == 0) {
      const bool p16 = g.flags & HFTT_SL_PRE_BF16;
      x3s_ln_rows<2>(yacc, prm + 192, prm + 256, hb, g.ln_mean, g.ln_rstd, tok, wave_ok, stage, j, lane,
                     (preb != nullptr && !p16) ? preb + (blk * 128 + wave * 32) * g.ldy : nullptr, ywave, g.ldy,
                     (preb != nullptr && p16) ? reinterpret_cast<unsigned short*>(preb) + (blk * 128 + wave * 32) * g.ldy : nullptr);
    }
  }
}

template <int E, int K32, int NT, bool LN, bool HR>
int launch_xs(const hftt_strip_desc& d, hipStream_t st) {
  using Cfg = XsCfg<K32, NT, LN>;
  static bool attr = false;
  if (!attr) { if (int rc = set_lds(x3s_linear_kernel<E, K32, NT, LN, HR>, Cfg::LDS, "x3s_strip_linear")) return rc; attr = true; }
  const int cus = n_cus();
  if (cus <= 0) { hftt_set_error("x3s_strip_linear: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  const long cap = (long)x3s_wgs(Cfg::LDS) * cus;
  hipLaunchKernelGGL((x3s_linear_kernel<E, K32, NT, LN, HR>), dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(256), Cfg::LDS, st, d);
  HFTT_CHECK_LAUNCH("x3s_strip_linear");
  return 0;
}
template <int MODE, bool HH>
int launch_xsm(const hftt_ffn_desc& d, hipStream_t st) {
  static bool attr = false;
  if (!attr) { if (int rc = set_lds(x3s_mlp_kernel<MODE, HH>, XSM_LDS, "x3s_strip_mlp")) return rc; attr = true; }
  const int cus = n_cus();
  if (cus <= 0) { hftt_set_error("x3s_strip_mlp: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((x3s_mlp_kernel<MODE, HH>), dim3((unsigned)(nblk < 2L * cus ? nblk : 2L * cus)), dim3(256), XSM_LDS, st, d);
  HFTT_CHECK_LAUNCH("x3s_strip_mlp");
  return 0;
}

// the shapes of the d = 64 model (K x N): forward 64x192 (q, k, v), 64x128 (cross k, v), 64x64 (cross q; fc_o + LayerNorm), backward 64x64
// (dX of fc_o, of the cross q), 192x64 / 128x64 (dX of the fused projections, + residual)
template <int E>
int dispatch_xs(const hftt_strip_desc& d, hipStream_t st) {
  const int k32 = d.K / 32, nt = d.N / 32;
  const bool hr = d.residual != nullptr;
  if (d.ln_gamma != nullptr) {
    if (k32 == 2 && nt == 2) return hr ? launch_xs<E, 2, 2, true, true>(d, st) : launch_xs<E, 2, 2, true, false>(d, st);
  } else if (k32 == 2) {
    if (nt == 2) return hr ? launch_xs<E, 2, 2, false, true>(d, st) : launch_xs<E, 2, 2, false, false>(d, st);
    if (nt == 4 && !hr) return launch_xs<E, 2, 4, false, false>(d, st);
    if (nt == 6 && !hr) return launch_xs<E, 2, 6, false, false>(d, st);
  } else if (nt == 2) {
    if (k32 == 4) return hr ? launch_xs<E, 4, 2, false, true>(d, st) : launch_xs<E, 4, 2, false, false>(d, st);
    if (k32 == 6) return hr ? launch_xs<E, 6, 2, false, true>(d, st) : launch_xs<E, 6, 2, false, false>(d, st);
  }
  hftt_set_error("x3s_strip_linear: shape N=%d K=%d%s is not covered (K x N in {64x64, 64x128, 64x192, 128x64, 192x64}; LayerNorm: 64x64)", d.N, d.K,
                 d.ln_gamma != nullptr ? " with LayerNorm" : "");
  return 1;
}
