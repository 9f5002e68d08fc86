// Fused multi-head attention of the split-operand ("x3") precision mode (gfx950): forward and backward, fp32 tensors in HBM,
// every product in three MFMA passes on 16-bit halves (x3_common.h).  Contract: include/hftt_hip.h (hftt_attn_fwd / hftt_attn_bwd, npass 2).
//
// Structure as in attn_fwd.hip / attn_bwd.hip (one workgroup per (sequence, head); the whole K / V of a head staged once in LDS;
// S^T = K.Q^T so a lane owns one query and its softmax row is register-local; accumulator tiles fed back as MFMA operands), with
//   * forward: Q, K, V and the probabilities P as fp16 hi / lo pairs (scores reach ~1e5 in the reference's first layer: 16 significant
//     bits are not enough there, 22 are);
//   * backward: the scores are recomputed in the SAME arithmetic and summation order as the forward (so P is the forward's, bit for
//     bit, from the stored row maximum and 1/sum); the four gradient products (dP = dO.V^T, dV = Pd^T.dO, dK = dS^T.Q, dQ = dS.K) take
//     bf16 hi / lo pairs: gradients of 1e-5 .. 1e-10 need fp32's exponent range.  Q and K therefore live in both forms.
#include <type_traits>
#include "hftt_common.h"
#include "x3_common.h"
#include "hftt_host.h"
#include "x3_internal.h"
#include "../../include/hftt_hip.h"
#include <math.h>

#ifdef HFTT_X3_ATTN_ABLATE
#define XABL(g, bit) (((g).pad & (bit)) != 0)
#else
#define XABL(g, bit) false
#endif

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <int KT, int DH>
struct XfCfg {
  static constexpr int LKP = KT * 32;
  static constexpr int RSK = DH + 8;                  // 144 B / 80 B rows: conflict-free ds_read_b128
  static constexpr int RSV = (DH == 64) ? ((KT == 8) ? 72 : 96) : 32;      // tr16 reads (see attn_fwd.hip)
  static constexpr int K_PL = LKP * RSK;              // elements of one K plane
  static constexpr int V_PL = LKP * RSV;
  static constexpr int LDS_BYTES = (2 * K_PL + 2 * V_PL) * 2 + 8 * 32 * 4;      // + one 1/sum per query row and wave (fused loop below)
};

// NW waves per workgroup, one 32-query block per wave at a time: 8 waves for the long key axes (KT = 8: K + V take 147 KB of LDS, so one
// workgroup per CU -- two waves per SIMD then let one wave's softmax run under the other's MFMAs, and the staging phase has twice the
// threads), 4 otherwise
// MAP: the attention map (g.probs) is an output -- the phase-wise form below; otherwise the fused per-tile loop
template <int KT, int DH, int NW, bool MAP>
__global__ __launch_bounds__(NW * 64) void x3_attn_fwd_kernel(const hftt_attn_desc g) {
  using Cfg = XfCfg<KT, DH>;
  constexpr int E = X3_F16;
  constexpr int NTHR = NW * 64;
  constexpr int RSK = Cfg::RSK, RSV = Cfg::RSV, LKP = Cfg::LKP, K_PL = Cfg::K_PL, V_PL = Cfg::V_PL;
  constexpr int KS = DH / 16, NT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* Ks = reinterpret_cast<unsigned short*>(smem);      // hi plane, then lo plane
  unsigned short* Vs = Ks + 2 * K_PL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int seq = blockIdx.x / g.n_heads, head = blockIdx.x % g.n_heads;
  const int Lq = g.Lq, Lk = g.Lk;

  // ---- stage K and V of this (seq, head): fp32 -> (hi, lo) planes, 8 + 8 x 16 B loads in flight per thread ----
  {
    const long kofs = (long)seq * g.k_seq_stride + head * DH;
    const long vofs = (long)seq * g.v_seq_stride + head * DH;
    constexpr int F4R = DH / 4;
    constexpr int TOTAL = LKP * F4R;
    constexpr int PER = (TOTAL + NTHR - 1) / NTHR;
    constexpr int UB = PER < 8 ? PER : 8;
    for (int base = 0; base < TOTAL; base += NTHR * UB) {
      float4 kf[UB], vf[UB];
#pragma unroll
      for (int u = 0; u < UB; u++) {
        const int i = base + tid + NTHR * u;
        const int ic = i < TOTAL ? i : TOTAL - 1;
        const int key = ic / F4R, c4 = ic % F4R;
        const int kc = key < Lk ? key : Lk - 1;       // clamped address: loads stay unconditional
        kf[u] = *reinterpret_cast<const float4*>(g.k + kofs + (long)kc * g.ldk + c4 * 4);
        vf[u] = *reinterpret_cast<const float4*>(g.v + vofs + (long)kc * g.ldv + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < UB; u++) {
        const int i = base + tid + NTHR * u;
        if (i >= TOTAL) continue;
        const int key = i / F4R, c4 = i % F4R;
        if (key >= Lk) { kf[u] = make_float4(0.f, 0.f, 0.f, 0.f); vf[u] = kf[u]; }
        uint2 hi, lo;
        x3_split4<E>(kf[u], hi, lo);
        *reinterpret_cast<uint2*>(Ks + key * RSK + c4 * 4) = hi;
        *reinterpret_cast<uint2*>(Ks + K_PL + key * RSK + c4 * 4) = lo;
        x3_split4<E>(vf[u], hi, lo);
        *reinterpret_cast<uint2*>(Vs + key * RSV + c4 * 4) = hi;
        *reinterpret_cast<uint2*>(Vs + V_PL + key * RSV + c4 * 4) = lo;
      }
    }
  }
  __syncthreads();

  const float scale = 1.0f / sqrtf((float)DH);
  const float c2 = scale * LOG2E;
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const int nqb = (Lq + 31) / 32;
  const bool vec_probs = (Lk % 4) == 0;
  const uint64_t hk = hftt_hash_key(g.drop_seed, g.drop_site);
  const bool quad_ok = (Lk & 3) == 0 && (((uint64_t)g.n_seq * (uint64_t)g.n_heads * (uint64_t)Lq * (uint64_t)Lk) >> 34) == 0;

  for (int qb = wave; qb < nqb; qb += NW) {
    const int qrow = qb * 32 + lr;                     // this lane's query (as the B-operand column)
    const int qrow_c = qrow < Lq ? qrow : Lq - 1;
    const float* qp = g.q + (long)seq * g.q_seq_stride + (long)qrow_c * g.ldq + head * DH + 8 * lh;
    bf16x8 qh[KS], ql[KS];                             // unscaled: the scale is folded into the exponent
#pragma unroll
    for (int s = 0; s < KS; s++) {
      const float4 f0 = *reinterpret_cast<const float4*>(qp + 16 * s);
      const float4 f1 = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
      const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
      x3_split8<E>(v, qh[s], ql[s]);
    }
    // ---- S^T = K . Q^T  (rows = keys in registers, column = this lane's query) ----
    f32x16 sacc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) sacc[kt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const unsigned short* p = Ks + (kt * 32 + lr) * RSK + 16 * s + 8 * lh;
        sacc[kt] = x3_mma<E>(lds_read_b128(p), lds_read_b128(p + K_PL), qh[s], ql[s], sacc[kt]);
      }
    // ---- softmax over keys: p = 2^((s - max) * c2), c2 = scale*log2e; the RAW maximum is subtracted exactly and stored (attn_fwd.hip) ----
    int lh4 = 4 * lh;
    asm volatile("" : "+v"(lh4));                      // opaque: keeps the per-register key numbers out of LICM's reach (spills)
    float mx = -INFINITY;
    if (Lk < LKP) {                                    // (wave-uniform) key padding of the last tile
      const int lkm = Lk - lh4;
#pragma unroll
      for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (kt * 32 + (r & 3) + 8 * (r >> 2) >= lkm) sacc[kt][r] = -INFINITY;
    }
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, sacc[kt][r]);
    mx = xor32_max(mx);
    if (!MAP) {
      // ---- no attention map wanted (every launch but the last decoder layer's cross-attention): ONE loop over the key tiles --
      // exponentials, row-sum contribution, dropout, fp16 split, then this tile's P.V MFMAs -- with the UNNORMALISED probabilities (<= 1,
      // the maximum element exactly 1); 1/sum multiplies the 32 x dh output rows at the end.  The exponentials / dropout / split of tile
      // kt + 1 then issue while the matrix pipe works on tile kt: in phases (all exponentials, all dropout, then P.V) the softmax's VALU
      // work ran with the matrix pipe idle.
      const long prow = (((long)seq * g.n_heads + head) * Lq + qrow) * (long)Lk;
      const uint32_t q0lo = (uint32_t)((uint64_t)prow >> 2) + (uint32_t)(lh4 >> 2);
      f32x16 oacc[NT];
#pragma unroll
      for (int n = 0; n < NT; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[n][r] = 0.f;
      // the dropout form is a compile-time parameter of the loop (DM: 0 none, 1 one hash per key quad, 2 per element): as run-time tests
      // inside the tile loop the forms met in phi copies of the whole tile and a branch per quad.  The kept probabilities enter P.V
      // unscaled; inv_keep rides on 1/sum at the end.
      auto pv_fused = [&](auto dm_c) __attribute__((always_inline)) {
        constexpr int DM = decltype(dm_c)::value;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
          float pt[16];
#pragma unroll
          for (int r = 0; r < 16; r++) {
            pt[r] = __builtin_amdgcn_exp2f((sacc[kt][r] - mx) * c2);
            sum += pt[r];
          }
          if (DM == 1) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
              const uint32_t w = hftt_hash_mix(hk, q0lo + (uint32_t)(kt * 8 + 2 * c), 0u);      // keys kt*32 + 8c + 4lh + {0..3}
              pt[4 * c] = (w & 0xFFu) < thr ? pt[4 * c] : 0.f;
              pt[4 * c + 1] = ((w >> 8) & 0xFFu) < thr ? pt[4 * c + 1] : 0.f;
              pt[4 * c + 2] = ((w >> 16) & 0xFFu) < thr ? pt[4 * c + 2] : 0.f;
              pt[4 * c + 3] = (w >> 24) < thr ? pt[4 * c + 3] : 0.f;
            }
          } else if (DM == 2) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
              const int key0 = kt * 32 + 8 * c + lh4;
#pragma unroll
              for (int e = 0; e < 4; e++)
                pt[4 * c + e] = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)(prow + key0 + e), thr) ? pt[4 * c + e] : 0.f;
            }
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; s2++) {
            bf16x8 ph, pl;
            x3_split8_nc<E>(pt + 8 * s2, ph, pl);
#pragma unroll
            for (int n = 0; n < NT; n++) {
              const int col = n * 32 + 16 * (gi & 1) + 4 * pp;
              const int r0 = kt * 32 + 16 * s2 + 4 * lh + qq;     // + 8 for the second half of the fragment
              const unsigned short* pvp = Vs + r0 * RSV + col;
              const bf16x8 vh = join4(lds_read_tr16(pvp), lds_read_tr16(pvp + 8 * RSV));
              const bf16x8 vl = join4(lds_read_tr16(pvp + V_PL), lds_read_tr16(pvp + V_PL + 8 * RSV));
              oacc[n] = x3_mma<E>(ph, pl, vh, vl, oacc[n]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);      // one tile at a time in program order (the MFMAs still run under the next tile's VALU work)
        }
        return sum;
      };
      float sum;
      if (g.drop_p > 0.f) {        // (wave-uniform)
        if (quad_ok) sum = pv_fused(std::integral_constant<int, 1>{}); else sum = pv_fused(std::integral_constant<int, 2>{});
      } else sum = pv_fused(std::integral_constant<int, 0>{});
      sum = xor32_sum(sum);
      const float inv = 1.0f / sum;
      if (lh == 0 && qrow < Lq) {
        float* st = g.lse + (((long)seq * g.n_heads + head) * Lq + qrow) * 2;
        st[0] = mx; st[1] = inv;
      }
      // the output tile has the QUERY on its register index (row acc_row32(r, lh)): 1/sum of those rows comes through 128 bytes of LDS
      float* invs = reinterpret_cast<float*>(smem + (size_t)(2 * K_PL + 2 * V_PL) * 2) + wave * 32;
      if (lh == 0) invs[lr] = inv * inv_keep;
      const long oofs = (long)seq * g.o_seq_stride + head * DH;
#pragma unroll
      for (int j4 = 0; j4 < 4; j4++) {
        const float4 i4 = *reinterpret_cast<const float4*>(invs + 8 * j4 + lh4);      // rows 8*j4 + 4*lh + {0..3} = acc_row32(4*j4 + e, lh)
        const float iv[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int q = qb * 32 + 8 * j4 + lh4 + e;
#pragma unroll
          for (int n = 0; n < NT; n++)
            if (q < Lq) g.out[oofs + (long)q * g.ldo + n * 32 + lr] = oacc[n][4 * j4 + e] * iv[e];
        }
      }
      continue;
    }
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f((sacc[kt][r] - mx) * c2);
        sacc[kt][r] = p;
        sum += p;
      }
    sum = xor32_sum(sum);
    const float inv = 1.0f / sum;
    if (lh == 0 && qrow < Lq) {
      float* st = g.lse + (((long)seq * g.n_heads + head) * Lq + qrow) * 2;
      st[0] = mx; st[1] = inv;
    }

    const long prow = (((long)seq * g.n_heads + head) * Lq + qrow) * (long)Lk;   // element index base of this query's row
    const float nrm = (g.probs != nullptr) ? inv_keep : inv * inv_keep;
    const uint32_t q0lo = (uint32_t)((uint64_t)prow >> 2) + (uint32_t)(lh4 >> 2);
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int key0 = kt * 32 + 8 * c + lh4;
        if (g.probs != nullptr) {    // (wave-uniform) the attention map is an output: normalise, store, then drop
          float p4[4];
#pragma unroll
          for (int e = 0; e < 4; e++) { p4[e] = sacc[kt][4 * c + e] * inv; sacc[kt][4 * c + e] = p4[e]; }
          if (qrow < Lq) {
            if (vec_probs && key0 + 3 < Lk) {
              *reinterpret_cast<float4*>(g.probs + prow + key0) = make_float4(p4[0], p4[1], p4[2], p4[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; e++)
                if (key0 + e < Lk) g.probs[prow + key0 + e] = p4[e];
            }
          }
        }
        if (g.drop_p > 0.f) {        // (wave-uniform)
          if (quad_ok) {
            const uint32_t w = hftt_hash_mix(hk, q0lo + (uint32_t)(kt * 8 + 2 * c), 0u);      // keys kt*32 + 8c + 4lh + {0..3}
            const float a0 = sacc[kt][4 * c], a1 = sacc[kt][4 * c + 1], a2 = sacc[kt][4 * c + 2], a3 = sacc[kt][4 * c + 3];
            sacc[kt][4 * c] = (w & 0xFFu) < thr ? a0 * nrm : 0.f;
            sacc[kt][4 * c + 1] = ((w >> 8) & 0xFFu) < thr ? a1 * nrm : 0.f;
            sacc[kt][4 * c + 2] = ((w >> 16) & 0xFFu) < thr ? a2 * nrm : 0.f;
            sacc[kt][4 * c + 3] = (w >> 24) < thr ? a3 * nrm : 0.f;
          } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
              sacc[kt][4 * c + e] = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)(prow + key0 + e), thr) ? sacc[kt][4 * c + e] * nrm : 0.f;
          }
        } else if (g.probs == nullptr) {
#pragma unroll
          for (int e = 0; e < 4; e++) sacc[kt][4 * c + e] *= inv;
        }
      }
    }
    // ---- out = P . V  (probability tile reused as the A operand; keys are the reduction index) ----
    f32x16 oacc[NT];
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) oacc[n][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) pv[e] = sacc[kt][8 * s2 + e];
        bf16x8 ph, pl;
        x3_split8<E>(pv, ph, pl);
#pragma unroll
        for (int n = 0; n < NT; n++) {
          const int col = n * 32 + 16 * (gi & 1) + 4 * pp;
          const int r0 = kt * 32 + 16 * s2 + 4 * lh + qq;     // + 8 for the second half of the fragment
          const unsigned short* p = Vs + r0 * RSV + col;
          const bf16x8 vh = join4(lds_read_tr16(p), lds_read_tr16(p + 8 * RSV));
          const bf16x8 vl = join4(lds_read_tr16(p + V_PL), lds_read_tr16(p + V_PL + 8 * RSV));
          oacc[n] = x3_mma<E>(ph, pl, vh, vl, oacc[n]);
        }
      }
    }
    const long oofs = (long)seq * g.o_seq_stride + head * DH;
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int q = qb * 32 + acc_row32(r, lh);
        if (q < Lq) g.out[oofs + (long)q * g.ldo + n * 32 + lr] = oacc[n][r];
      }
  }
}

template <int KT, int DH, bool MAP>
int launch_xf2(const hftt_attn_desc& d, hipStream_t st) {
  using Cfg = XfCfg<KT, DH>;
  constexpr int NW = (KT == 8) ? 8 : 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(x3_attn_fwd_kernel<KT, DH, NW, MAP>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) { hftt_set_error("x3_attn_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  hipLaunchKernelGGL((x3_attn_fwd_kernel<KT, DH, NW, MAP>), dim3((unsigned)(d.n_seq * d.n_heads)), dim3(NW * 64), Cfg::LDS_BYTES, st, d);
  HFTT_CHECK_LAUNCH("x3_attn_fwd");
  return 0;
}
template <int KT, int DH>
int launch_xf(const hftt_attn_desc& d, hipStream_t st) {
  return d.probs != nullptr ? launch_xf2<KT, DH, true>(d, st) : launch_xf2<KT, DH, false>(d, st);
}
template <int DH>
int dispatch_xf(const hftt_attn_desc& d, hipStream_t st) {
  const int kt = (d.Lk + 31) / 32;
  if (kt <= 1) return launch_xf<1, DH>(d, st);
  if (kt <= 2) return launch_xf<2, DH>(d, st);
  if (kt <= 3) return launch_xf<3, DH>(d, st);
  if (kt <= 4) return launch_xf<4, DH>(d, st);
  return launch_xf<8, DH>(d, st);
}

#include "x3_attn_bwd.h"

}  // namespace

int hftt_x3_attn_fwd(const hftt_attn_desc& d, hipStream_t st) {
  const int rcp = hftt_x3p_attn_fwd_try(d, st);      // q, k, v as f16-pair planes: the LDS-DMA form (x3_attn_pl.hip)
  if (rcp >= 0) return rcp;
  return d.dh == 64 ? dispatch_xf<64>(d, st) : dispatch_xf<32>(d, st);
}
// Ablation build only (tools/ablate_x3_attn.sh, -DHFTT_X3_ATTN_ABLATE): HFTT_X3_ATTN_DEBUG switches mechanisms of the backward off
// (timing experiments, results garbage): 1 no Q / dO staging conversions, 2 no S / dP MFMAs,
// 4 no softmax-backward arithmetic, 8 no dV / dK products (splits + MFMAs), 16 no dS -> LDS and no dQ product, 32 no dQ stores
static int x3_attn_debug() { static const int v = [] { const char* e = getenv("HFTT_X3_ATTN_DEBUG"); return e ? atoi(e) : 0; }(); return v; }
int hftt_x3_attn_bwd(const hftt_attn_desc& d0, hipStream_t st) {
  hftt_attn_desc d = d0;
#ifdef HFTT_X3_ATTN_ABLATE
  d.pad = (uint32_t)x3_attn_debug();
#else
  d.pad = 0;
#endif
  const unsigned both = HFTT_ATTN_Q_F16PAIR | HFTT_ATTN_KV_F16PAIR;
  if (d.io_flags & both) return hftt_x3p_attn_bwd(d, st);      // q, k, v as f16-pair planes: x3_attn_pl.hip
  return d.dh == 64 ? dispatch_xb<64, false>(d, st) : dispatch_xb<32, false>(d, st);
}
