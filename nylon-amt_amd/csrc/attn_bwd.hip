// Fused multi-head attention backward (gfx950), one workgroup per (sequence, head), one wave per 32 keys.
//   recompute P = exp(Q K^T/sqrt(dh) - rowmax) * inv_rowsum (the forward's row statistics => the forward's P);  dP = dO V^T;  dS = P*(dP - rowsum(dO*O))/sqrt(dh)
//   dV = Pd^T dO,  dK = dS^T Q,  dQ = dS K                                  -- include/hftt_hip.h (hftt_attn_bwd)
// Layout choices (all lane maps verified by tools/probe_mfma):
//   * S and dP are produced with the KEY on the MFMA lane (B operand = this wave's K / V rows, held in registers for
//     the whole kernel), so P and dS accumulator tiles are directly the B operands of dV^T += dO^T.P and
//     dK^T += Q^T.dS (accumulator-as-operand, probe T4 "Y = A2*X"); dK^T/dV^T never leave the wave's registers.
//   * only dS crosses LDS (bf16 [query][key] image) for dQ = dS.K, computed with 16x16x32 tiles spread over the waves;
//     K for that product is the row-major LDS image read through ds_read_b64_tr_b16.
//   * Q and dO blocks (32 rows) are staged per iteration; their transposes are again tr reads of the same image.
#include <type_traits>
#include "hftt_common.h"
#include "hftt_host.h"
#include "x3_internal.h"
#include "../../include/hftt_hip.h"
#include <math.h>

int hftt_attn_check(const hftt_attn_desc* d, bool bwd);

// Ablation build (tools/ablate_attn.sh, -DHFTT_ATTN_ABLATE): descriptor pad bits switch mechanisms off (timing only, results garbage):
//   1 no Q/dO/O block loads (registers keep the first block)   2 no S / dP MFMAs   4 no softmax-backward arithmetic   8 no dropout
//   16 no dV / dK MFMAs   32 no dS -> LDS and no dQ product / stores   64 no dK / dV stores
#ifdef HFTT_ATTN_ABLATE
#define ABL(g, bit) (((g).pad & (bit)) != 0)
#else
#define ABL(g, bit) false
#endif

namespace {

// npass == 1: bf16 planes as described above.  npass == 3 (parity): every tile stays fp32 in LDS / registers and all
// products run on v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (one f32 per lane per operand; the accumulator tiles of P and dS are
// fed back as B operands register-for-register).
template <int KT, int DH, int NPASS>
struct AbCfg {
  static constexpr bool F32 = (NPASS == 3);
  static constexpr int LKP = KT * 32;
  static constexpr int RSK = F32 ? DH + 1 : ((DH == 64) ? 96 : 32);   // bf16: tr-read friendly (64 mod 128 bytes)
  static constexpr int RSQ = F32 ? DH + 1 : DH + 8;                   // bf16: b128-read friendly
  static constexpr int RSS = F32 ? LKP + 1 : LKP + 8;
  static constexpr int K_ELEMS = LKP * RSK;
  static constexpr int Q_ELEMS = 32 * RSQ;
  static constexpr int S_ELEMS = 32 * RSS;
  static constexpr int ELEMS = K_ELEMS + 2 * Q_ELEMS + S_ELEMS;
  static constexpr int LDS_BYTES = ELEMS * (F32 ? 4 : 2) + 96 * 4;
  static constexpr int NTHR = KT * 64;
};

// HB: q, k, v, out, dout all stored as bf16 -> 16-byte loads straight into LDS / fragments (dq, dk, dv dtype by io_flags)
template <int KT, int DH, int NPASS, bool HB>
#ifdef HFTT_ATTN_BWD_NOCAP      // (A/B build of tools/r05_step23.sh)
#define HFTT_AB_WAVES(NPASS, KT, HB) 1
#else
#define HFTT_AB_WAVES(NPASS, KT, HB) (((NPASS) == 1 && ((KT) == 4 || ((KT) == 3 && (HB)))) ? 2 : 1)
#endif
__global__ __launch_bounds__(KT * 64, HFTT_AB_WAVES(NPASS, KT, HB)) void attn_bwd_kernel(const hftt_attn_desc g) {   // (see x3_attn_bwd.h)
  using Cfg = AbCfg<KT, DH, NPASS>;
  constexpr bool F32 = Cfg::F32;
  static_assert(!(F32 && HB), "bf16-stored tensors are a bf16-mode feature");
  constexpr int RSK = Cfg::RSK, RSQ = Cfg::RSQ, RSS = Cfg::RSS, LKP = Cfg::LKP, NTHR = Cfg::NTHR;
  constexpr int KS = DH / 16, NT = DH / 32, F4R = DH / 4, HD = DH / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* Ks16 = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Qs16 = Ks16 + Cfg::K_ELEMS;
  unsigned short* Os16 = Qs16 + Cfg::Q_ELEMS;
  unsigned short* Ss16 = Os16 + Cfg::Q_ELEMS;
  float* Ks32 = reinterpret_cast<float*>(smem);
  float* Qs32 = Ks32 + Cfg::K_ELEMS;
  float* Os32 = Qs32 + Cfg::Q_ELEMS;
  float* Ss32 = Os32 + Cfg::Q_ELEMS;
  float* lse_s = reinterpret_cast<float*>(smem + (size_t)Cfg::ELEMS * (F32 ? 4 : 2));
  float* delta_s = lse_s + 32;
  float* inv_s = lse_s + 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int seq = blockIdx.x / g.n_heads, head = blockIdx.x % g.n_heads;
  const int Lq = g.Lq, Lk = g.Lk;
  const float scale = 1.0f / sqrtf((float)DH);
  constexpr float LOG2E = 1.4426950408889634f;
  const float c2 = scale * LOG2E;
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  // one hash per four elements needs Lk % 4 == 0 and quad indices below 2^32 for the whole launch (attn_fwd.hip makes the same choice;
  // either way the decisions are those of hftt_keep)
  const bool pair_ok = (Lk & 3) == 0 && (((uint64_t)g.n_seq * (uint64_t)g.n_heads * (uint64_t)g.Lq * (uint64_t)Lk) >> 34) == 0;

  auto pack4 = [&](const float4& f) {
    uint2 ph;
    ph.x = f2bf(f.x) | ((unsigned)f2bf(f.y) << 16); ph.y = f2bf(f.z) | ((unsigned)f2bf(f.w) << 16);
    return ph;
  };
  auto put_row4 = [&](float* dst, const float4& f) { dst[0] = f.x; dst[1] = f.y; dst[2] = f.z; dst[3] = f.w; };

  const bool q_bf = !F32 && (g.io_flags & HFTT_ATTN_Q_BF16), kv_bf = !F32 && (g.io_flags & HFTT_ATTN_KV_BF16), o_bf = !F32 && (g.io_flags & HFTT_ATTN_O_BF16);
  const bool dq_bf = !F32 && (g.io_flags & HFTT_ATTN_DQ_BF16), dkv_bf = !F32 && (g.io_flags & HFTT_ATTN_DKV_BF16);
  const long kofs = (long)seq * g.k_seq_stride + head * DH;      // element offsets (tensors may be fp32 or bf16)
  const long vofs = (long)seq * g.v_seq_stride + head * DH;
  const float* kb = g.k + kofs;                                  // pointer forms: fp32 (parity) path only
  const float* vb = g.v + vofs;
  // ---- stage all of K (row-major) for the dQ product: all of a thread's loads in flight, then convert + store ----
  if (HB) {
    constexpr int S8R = DH / 8;
    constexpr int KCNT8 = (LKP * S8R + NTHR - 1) / NTHR;
    const unsigned short* kp = reinterpret_cast<const unsigned short*>(g.k) + kofs;
    uint4 kst[KCNT8];
#pragma unroll
    for (int u = 0; u < KCNT8; u++) {
      const int i = tid + NTHR * u;
      const int ic = i < LKP * S8R ? i : LKP * S8R - 1;
      const int key = ic / S8R, c8 = ic % S8R;
      const int kc = key < Lk ? key : Lk - 1;
      kst[u] = *reinterpret_cast<const uint4*>(kp + (long)kc * g.ldk + c8 * 8);
      if (key >= Lk) kst[u] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < KCNT8; u++) {
      const int i = tid + NTHR * u;
      if (i < LKP * S8R) *reinterpret_cast<uint4*>(Ks16 + (i / S8R) * RSK + (i % S8R) * 8) = kst[u];
    }
  } else {
    constexpr int KCNT = (LKP * F4R + NTHR - 1) / NTHR;
    float4 kst[KCNT];
#pragma unroll
    for (int u = 0; u < KCNT; u++) {
      const int i = tid + NTHR * u;
      const int ic = i < LKP * F4R ? i : LKP * F4R - 1;
      const int key = ic / F4R, c4 = ic % F4R;
      const int kc = key < Lk ? key : Lk - 1;       // clamped address + select: loads stay unconditional
      kst[u] = hftt_load4(g.k, kv_bf, kofs + (long)kc * g.ldk + c4 * 4);
      if (key >= Lk) kst[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < KCNT; u++) {
      const int i = tid + NTHR * u;
      if (i < LKP * F4R) {
        const int key = i / F4R, c4 = i % F4R;
        if (F32) put_row4(Ks32 + key * RSK + c4 * 4, kst[u]);
        else *reinterpret_cast<uint2*>(Ks16 + key * RSK + c4 * 4) = pack4(kst[u]);
      }
    }
  }
  // ---- bf16 forms: the dQ image holds K - mean_key(K).  dQ_i = sum_j dS_ij k_j and sum_j dS_ij = 0 (softmax rows sum to one), so any common
  //      vector may be taken off the keys -- in exact arithmetic.  With dS rounded to bf16 (and delta formed from the bf16-stored O) the row
  //      sums are only ~2^-9 |dS|, and that residue times the COMMON part of the keys swamped the signal wherever the keys of a sequence are
  //      nearly alike (the decoder's self-attention over time: cos(dQ, fp64) = 0.1 .. 0.4); with the mean removed it multiplies only the
  //      spread of the keys.  The S recompute keeps the raw K (registers, below): the forward's P exactly. ----
  if (!F32) {
    constexpr int G = NTHR / DH;                       // row groups: thread = (column c, group gq)
    static_assert(NTHR % DH == 0 && G >= 1, "one thread per (column, row group)");
    float* part = reinterpret_cast<float*>(Ss16);      // [G][DH] partial sums (the dS image is not live before stage (g))
    __syncthreads();
    const int c = tid % DH, gq = tid / DH;
    float cs = 0.f;
    for (int row = gq; row < Lk; row += G) cs += bf2f(Ks16[row * RSK + c]);
    part[gq * DH + c] = cs;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int u = 0; u < G; u++) mean += part[u * DH + c];
    mean /= (float)Lk;
    for (int row = gq; row < Lk; row += G) Ks16[row * RSK + c] = (unsigned short)f2bf(bf2f(Ks16[row * RSK + c]) - mean);
    // (ordered before stage (i) by the barriers of the query-block loop; `part` is overwritten by stage (g) only after barrier (b))
  }
  // ---- this wave's K and V rows as B-operand fragments (B[k = dh][col = key]) ----
  const int mykey = wave * 32 + lr;
  const int mykey_c = mykey < Lk ? mykey : Lk - 1;     // clamped address for unconditional loads
  bf16x8 kfh[KS], vfh[KS];        // bf16 path
  float kf32[HD], vf32[HD];       // fp32 path: K/V[key][HD*lh + t]
  if (F32) {
#pragma unroll
    for (int t4 = 0; t4 < HD / 4; t4++) {
      float4 a = *reinterpret_cast<const float4*>(kb + (long)mykey_c * g.ldk + HD * lh + 4 * t4);
      float4 b = *reinterpret_cast<const float4*>(vb + (long)mykey_c * g.ldv + HD * lh + 4 * t4);
      if (mykey >= Lk) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
      kf32[4 * t4] = a.x; kf32[4 * t4 + 1] = a.y; kf32[4 * t4 + 2] = a.z; kf32[4 * t4 + 3] = a.w;
      vf32[4 * t4] = b.x; vf32[4 * t4 + 1] = b.y; vf32[4 * t4 + 2] = b.z; vf32[4 * t4 + 3] = b.w;
    }
  } else if (HB) {
    const unsigned short* kp = reinterpret_cast<const unsigned short*>(g.k) + kofs + (long)mykey_c * g.ldk;
    const unsigned short* vp = reinterpret_cast<const unsigned short*>(g.v) + vofs + (long)mykey_c * g.ldv;
#pragma unroll
    for (int s = 0; s < KS; s++) {
      kfh[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s + 8 * lh);
      vfh[s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s + 8 * lh);
      if (mykey >= Lk) {
#pragma unroll
        for (int e = 0; e < 8; e++) { kfh[s][e] = 0; vfh[s][e] = 0; }
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < KS; s++) {
      float kv[8], vv[8];
      {
        const float4 a0 = hftt_load4(g.k, kv_bf, kofs + (long)mykey_c * g.ldk + 16 * s + 8 * lh);
        const float4 a1 = hftt_load4(g.k, kv_bf, kofs + (long)mykey_c * g.ldk + 16 * s + 8 * lh + 4);
        const float4 b0 = hftt_load4(g.v, kv_bf, vofs + (long)mykey_c * g.ldv + 16 * s + 8 * lh);
        const float4 b1 = hftt_load4(g.v, kv_bf, vofs + (long)mykey_c * g.ldv + 16 * s + 8 * lh + 4);
        kv[0] = a0.x; kv[1] = a0.y; kv[2] = a0.z; kv[3] = a0.w; kv[4] = a1.x; kv[5] = a1.y; kv[6] = a1.z; kv[7] = a1.w;
        vv[0] = b0.x; vv[1] = b0.y; vv[2] = b0.z; vv[3] = b0.w; vv[4] = b1.x; vv[5] = b1.y; vv[6] = b1.z; vv[7] = b1.w;
      }
      if (mykey >= Lk) {
#pragma unroll
        for (int e = 0; e < 8; e++) { kv[e] = 0.f; vv[e] = 0.f; }
      }
#pragma unroll
      for (int e = 0; e < 8; e++) { kfh[s][e] = (short)f2bf(kv[e]); vfh[s][e] = (short)f2bf(vv[e]); }
    }
  }
  f32x16 dKT[NT], dVT[NT];
#pragma unroll
  for (int n = 0; n < NT; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dKT[n][r] = 0.f; dVT[n][r] = 0.f; }

  const long qofs = (long)seq * g.q_seq_stride + head * DH;
  const long oofs = (long)seq * g.o_seq_stride + head * DH;
  const long dqofs = (long)seq * g.dq_seq_stride + head * DH;
  const long sh = (long)seq * g.n_heads + head;
  const int nqb = (Lq + 31) / 32;

  // Q / dO / O rows (+ softmax row statistics) of the NEXT query block are fetched into registers while the current block is
  // being processed, so their HBM latency is off the critical path
  constexpr int QSLOT = HB ? DH / 8 : F4R;          // 16-byte slots per row (8 bf16 or 4 fp32)
  constexpr int QCNT = (32 * QSLOT + NTHR - 1) / NTHR;
  uint4 pq[QCNT], pdo[QCNT], po[QCNT];             // raw 16-byte slots
  float2 pl[QCNT];
  auto qload = [&](int qb) {
#pragma unroll
    for (int u = 0; u < QCNT; u++) {
      const int i = tid + NTHR * u;
      const int ic = i < 32 * QSLOT ? i : 32 * QSLOT - 1;
      const int row = ic / QSLOT, cs = ic % QSLOT;
      const int q = qb * 32 + row;
      const int qc = q < Lq ? q : Lq - 1;          // clamped address + select: loads stay unconditional
      if (HB) {
        pq[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(g.q) + qofs + (long)qc * g.ldq + cs * 8);
        pdo[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(g.dout) + oofs + (long)qc * g.ldo + cs * 8);
        po[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(g.out) + oofs + (long)qc * g.ldo + cs * 8);
      } else {
        const float4 a = hftt_load4(g.q, q_bf, qofs + (long)qc * g.ldq + cs * 4);
        const float4 b2 = hftt_load4(g.dout, o_bf, oofs + (long)qc * g.ldo + cs * 4);
        const float4 c = hftt_load4(g.out, o_bf, oofs + (long)qc * g.ldo + cs * 4);
        pq[u] = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w));
        pdo[u] = make_uint4(__float_as_uint(b2.x), __float_as_uint(b2.y), __float_as_uint(b2.z), __float_as_uint(b2.w));
        po[u] = make_uint4(__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), __float_as_uint(c.w));
      }
      pl[u] = *reinterpret_cast<const float2*>(g.lse + (sh * Lq + qc) * 2);
      // rows past Lq are zeroed when the registers are consumed (stage (a) of the next block): a select here would make hipcc
      // wait for each load right after issuing it, and the whole prefetch would sit on the critical path
    }
  };
  qload(0);

  for (int qb = 0; qb < nqb; qb++) {
    // ---- (a) registers -> LDS: the Q and dO blocks, delta = rowsum(dO*O), row statistics ----
#pragma unroll
    for (int u = 0; u < QCNT; u++) {
      const int i = tid + NTHR * u;
      if (i < 32 * QSLOT) {                         // wave-uniform (32*QSLOT and NTHR are multiples of 64)
        const int row = i / QSLOT, cs = i % QSLOT;
        if (qb * 32 + row >= Lq) { pq[u] = make_uint4(0u, 0u, 0u, 0u); pdo[u] = pq[u]; po[u] = pq[u]; pl[u] = make_float2(0.f, 0.f); }
        float dot;
        if (HB) {
          *reinterpret_cast<uint4*>(Qs16 + row * RSQ + cs * 8) = pq[u];
          *reinterpret_cast<uint4*>(Os16 + row * RSQ + cs * 8) = pdo[u];
          const unsigned dw[4] = {pdo[u].x, pdo[u].y, pdo[u].z, pdo[u].w};
          const unsigned ow[4] = {po[u].x, po[u].y, po[u].z, po[u].w};
          dot = 0.f;
#pragma unroll
          for (int e = 0; e < 4; e++) dot += bf2f(dw[e] & 0xFFFFu) * bf2f(ow[e] & 0xFFFFu) + bf2f(dw[e] >> 16) * bf2f(ow[e] >> 16);
        } else {
          const float4 qf = make_float4(__uint_as_float(pq[u].x), __uint_as_float(pq[u].y), __uint_as_float(pq[u].z), __uint_as_float(pq[u].w));
          const float4 df = make_float4(__uint_as_float(pdo[u].x), __uint_as_float(pdo[u].y), __uint_as_float(pdo[u].z), __uint_as_float(pdo[u].w));
          const float4 of = make_float4(__uint_as_float(po[u].x), __uint_as_float(po[u].y), __uint_as_float(po[u].z), __uint_as_float(po[u].w));
          if (F32) {
            put_row4(Qs32 + row * RSQ + cs * 4, qf);
            put_row4(Os32 + row * RSQ + cs * 4, df);
          } else {
            *reinterpret_cast<uint2*>(Qs16 + row * RSQ + cs * 4) = pack4(qf);
            *reinterpret_cast<uint2*>(Os16 + row * RSQ + cs * 4) = pack4(df);
          }
          dot = df.x * of.x + df.y * of.y + df.z * of.z + df.w * of.w;
        }
        dot = group_sum<QSLOT>(dot);
        if (cs == 0) {
          delta_s[row] = dot;
          lse_s[row] = pl[u].x;                             // HB: the RAW row maximum (attn_fwd.hip); else the maximum of the scaled scores
          inv_s[row] = pl[u].y;
        }
      }
    }
    __syncthreads();   // (b)
    if (qb + 1 < nqb && !ABL(g, 1)) qload(qb + 1);

    // ---- (c) S tile and (d) dP tile: rows = queries (registers), column = this lane's key ----
    f32x16 sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; r++) { sacc[r] = 0.f; pacc[r] = 0.f; }
    if (F32) {
#pragma unroll
      for (int t = 0; t < HD; t++) {
        sacc = mfma32_f32(Qs32[lr * RSQ + HD * lh + t], kf32[t], sacc);
        pacc = mfma32_f32(Os32[lr * RSQ + HD * lh + t], vf32[t], pacc);
      }
    } else if (!ABL(g, 2)) {
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const int off = lr * RSQ + 16 * s + 8 * lh;
        sacc = mfma32(lds_read_b128(Qs16 + off), kfh[s], sacc);
        pacc = mfma32(lds_read_b128(Os16 + off), vfh[s], pacc);
      }
    }
    const bool key_ok = mykey < Lk;
    // The 16 registers of a lane are 16 query rows: rows 8j + 4*lh + {0,1,2,3} for j = r >> 2 -> one 16-byte LDS read per statistic and j.
    // The (wave-uniform) dropout test is hoisted out of the register loop: 16 branches in it kept hipcc from interleaving the rows.
    auto softmax_bwd = [&](auto drop_c, auto pair_c) {
      constexpr bool DROP = decltype(drop_c)::value, PAIR = decltype(pair_c)::value;
      const long row0 = sh * Lq + (long)qb * 32 + 4 * lh;                  // element row of register 0
      const uint64_t ebase = (uint64_t)(row0 * (long)Lk + mykey);
      // PAIR (quad form): lanes 4i .. 4i+3 hold the four keys of one hash quad and registers 4j .. 4j+3 four adjacent rows.  Lane 4i+a hashes
      // the quad of row(4j + a); a DPP quad broadcast hands every lane each row's word, of which it takes its own key's byte: one hash per
      // FOUR elements.
      const int sub = lane & 3;
      const uint32_t fsh = 8u * (uint32_t)sub;
      const uint64_t hk = hftt_hash_key(g.drop_seed, g.drop_site);
      const uint32_t quarter = (uint32_t)(Lk >> 2);
      const uint32_t qlo = (uint32_t)(row0 + sub) * quarter + (uint32_t)(mykey >> 2);          // quad index of (row(0) + sub, my key quad): < 2^32 (pair_ok)
#pragma unroll
      for (int j4 = 0; j4 < 4; j4++) {
        const float4 m4 = *reinterpret_cast<const float4*>(lse_s + 8 * j4 + 4 * lh);
        const float4 i4 = *reinterpret_cast<const float4*>(inv_s + 8 * j4 + 4 * lh);
        const float4 d4 = *reinterpret_cast<const float4*>(delta_s + 8 * j4 + 4 * lh);
        const float rs_m[4] = {m4.x, m4.y, m4.z, m4.w}, rs_i[4] = {i4.x, i4.y, i4.z, i4.w}, rs_d[4] = {d4.x, d4.y, d4.z, d4.w};
        float m_[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP && PAIR) {
          const uint32_t w = hftt_hash_mix(hk, qlo + (uint32_t)(8 * j4) * quarter, 0u);      // row(4*j4) - row(0) = 8*j4 rows
          const uint32_t w0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x00, 0xF, 0xF, true);   // quad_perm [0,0,0,0]: row(4j+0)'s word
          const uint32_t w1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x55, 0xF, 0xF, true);   // [1,1,1,1]
          const uint32_t w2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xAA, 0xF, 0xF, true);   // [2,2,2,2]
          const uint32_t w3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xFF, 0xF, 0xF, true);   // [3,3,3,3]
          m_[0] = ((w0 >> fsh) & 0xFFu) < thr ? inv_keep : 0.f;
          m_[1] = ((w1 >> fsh) & 0xFFu) < thr ? inv_keep : 0.f;
          m_[2] = ((w2 >> fsh) & 0xFFu) < thr ? inv_keep : 0.f;
          m_[3] = ((w3 >> fsh) & 0xFFu) < thr ? inv_keep : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int r = 4 * j4 + e;
          float p;
          if (HB) p = __builtin_amdgcn_exp2f((sacc[r] - rs_m[e]) * c2) * rs_i[e];
          else {
            const float arg = hftt_attn_scaled<DH, F32>(sacc[r]) - rs_m[e];      // the forward's own scaled score (attn_fwd.hip)
            p = (F32 ? expf(arg) : __expf(arg)) * rs_i[e];
          }
          if (!key_ok) p = 0.f;
          float pd = p, dp = pacc[r];
          if (DROP) {
            float m;
            if (PAIR) m = m_[e];
            else m = hftt_keep(g.drop_seed, g.drop_site, ebase + (uint64_t)((e + 8 * j4) * Lk), thr) ? inv_keep : 0.f;
            pd = p * m;
            dp = dp * m;
          }
          sacc[r] = pd;                                        // dropped probabilities (for dV)
          pacc[r] = p * (dp - rs_d[e]) * scale;                // dS (scaled): for dK, dQ
        }
      }
    };
    if (ABL(g, 4)) {
    } else if (g.drop_p > 0.f && !ABL(g, 8)) {
      if (pair_ok) softmax_bwd(std::true_type{}, std::true_type{}); else softmax_bwd(std::true_type{}, std::false_type{});
    } else softmax_bwd(std::false_type{}, std::false_type{});
    // ---- (e) dV^T += dO^T . Pd   (f) dK^T += Q^T . dS ----
    if (F32) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int qrow = acc_row32(r, lh);
#pragma unroll
        for (int n = 0; n < NT; n++) {
          dVT[n] = mfma32_f32(Os32[qrow * RSQ + n * 32 + lr], sacc[r], dVT[n]);
          dKT[n] = mfma32_f32(Qs32[qrow * RSQ + n * 32 + lr], pacc[r], dKT[n]);
        }
      }
    } else if (!ABL(g, 16)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        bf16x8 ph, sh_;
#pragma unroll
        for (int e = 0; e < 8; e++) { ph[e] = (short)f2bf(sacc[8 * s2 + e]); sh_[e] = (short)f2bf(pacc[8 * s2 + e]); }
#pragma unroll
        for (int n = 0; n < NT; n++) {
          const int col = n * 32 + 16 * (gi & 1) + 4 * pp;
          const int r0 = 16 * s2 + 4 * lh + qq;
          const bf16x8 oh = join4(lds_read_tr16(Os16 + r0 * RSQ + col), lds_read_tr16(Os16 + (r0 + 8) * RSQ + col));
          dVT[n] = mfma32(oh, ph, dVT[n]);
          const bf16x8 qh = join4(lds_read_tr16(Qs16 + r0 * RSQ + col), lds_read_tr16(Qs16 + (r0 + 8) * RSQ + col));
          dKT[n] = mfma32(qh, sh_, dKT[n]);
        }
      }
    }
    // ---- (g) dS -> LDS [query][key] ----
    if (ABL(g, 32)) {
    } else if (F32) {
#pragma unroll
      for (int r = 0; r < 16; r++) Ss32[acc_row32(r, lh) * RSS + wave * 32 + lr] = pacc[r];
    } else {
      // lanes 2i / 2i+1 hold adjacent keys: one packed pair (ds_write_b32) per register pair instead of two 2-byte writes
      const bool odd = lane & 1;
#pragma unroll
      for (int rp = 0; rp < 8; rp++) {
        const unsigned pk = pair_rows_to_cols(pacc[2 * rp], pacc[2 * rp + 1], odd);
        *reinterpret_cast<unsigned*>(Ss16 + acc_row32(2 * rp + (odd ? 1 : 0), lh) * RSS + wave * 32 + (lr & ~1)) = pk;
      }
    }
    __syncthreads();   // (h)

    // ---- (i) dQ block = dS . K with 16x16 tiles spread over the waves ----
    constexpr int CT = DH / 16;
    for (int t = wave; t < 2 * CT && !ABL(g, 32); t += KT) {
      const int qh2 = t / CT, ct = t % CT;
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
      if (F32) {
#pragma unroll 8
        for (int k4 = 0; k4 < LKP / 4; k4++)
          a4 = mfma16_f32(Ss32[(qh2 * 16 + (lane & 15)) * RSS + k4 * 4 + gi], Ks32[(k4 * 4 + gi) * RSK + ct * 16 + (lane & 15)], a4);
      } else {
#pragma unroll 2
        for (int ks = 0; ks < KT; ks++) {
          const bf16x8 ah = lds_read_b128(Ss16 + (qh2 * 16 + (lane & 15)) * RSS + ks * 32 + 8 * gi);
          const int krow = ks * 32 + 8 * gi + qq;
          const int kcol = ct * 16 + 4 * pp;
          const bf16x8 bh = join4(lds_read_tr16(Ks16 + krow * RSK + kcol), lds_read_tr16(Ks16 + (krow + 4) * RSK + kcol));
          a4 = mfma16(ah, bh, a4);
        }
      }
      if (dq_bf) {
        // lanes 2i / 2i+1 hold adjacent columns: one packed 4-byte store per register pair
        unsigned short* dqp = reinterpret_cast<unsigned short*>(g.dq) + dqofs;
        const bool odd = lane & 1;
#pragma unroll
        for (int rp = 0; rp < 2; rp++) {
          const unsigned pk = pair_rows_to_cols(a4[2 * rp], a4[2 * rp + 1], odd);
          const int q = qb * 32 + qh2 * 16 + gi * 4 + 2 * rp + (odd ? 1 : 0);
          if (q < Lq) *reinterpret_cast<unsigned*>(dqp + (long)q * g.lddq + ct * 16 + ((lane & 15) & ~1)) = pk;
        }
      } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int q = qb * 32 + qh2 * 16 + gi * 4 + r;
        if (q < Lq) g.dq[dqofs + (long)q * g.lddq + ct * 16 + (lane & 15)] = a4[r];
      }
      }
    }
    // no barrier needed here: the next iteration's staging touches only Qs/Os/lse/delta, which no wave reads in (i);
    // barrier (b) of the next iteration orders (i) before the next (g).
  }

  // ---- epilogue: dK, dV (this wave's 32 keys) ----
  if (!F32 && dkv_bf) {
    // bf16 gradients leave through LDS: the accumulators hold, per lane, 4 consecutive head-dim elements of ONE key row, so a direct store
    // is 8 bytes per lane into 32 different rows per instruction.  Written row-major into the wave's own LDS patch and read back as 16 bytes
    // per lane, eight lanes cover the 128-byte row segment of this head: whole sectors per instruction (the scattered form wrote up to 2.7x
    // the bytes to HBM on the [S, 2d] cross-attention layout, whose 1 KB row pitch defeats the L2's write combining).
    constexpr int RSE = DH + 8;
    __syncthreads();                                           // every wave is done with the K / Q / dO / dS images
    unsigned short* ek = reinterpret_cast<unsigned short*>(smem) + wave * (2 * 32 * RSE);
    unsigned short* ev = ek + 32 * RSE;
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int dh0 = n * 32 + 8 * c + 4 * lh;
        *reinterpret_cast<uint2*>(ek + lr * RSE + dh0) = pack4(make_float4(dKT[n][4 * c], dKT[n][4 * c + 1], dKT[n][4 * c + 2], dKT[n][4 * c + 3]));
        *reinterpret_cast<uint2*>(ev + lr * RSE + dh0) = pack4(make_float4(dVT[n][4 * c], dVT[n][4 * c + 1], dVT[n][4 * c + 2], dVT[n][4 * c + 3]));
      }
    __syncthreads();
    constexpr int CPR = DH / 8;                                // 16-byte chunks per row
    constexpr int RPP = 64 / CPR;                              // rows per pass of the wave
    unsigned short* dkp = reinterpret_cast<unsigned short*>(g.dk) + (long)seq * g.dk_seq_stride + head * DH;
    unsigned short* dvp = reinterpret_cast<unsigned short*>(g.dv) + (long)seq * g.dv_seq_stride + head * DH;
    const bool wide_ok = !ABL(g, 64) && (g.lddk % 8 == 0) && (g.lddv % 8 == 0) && (g.dk_seq_stride % 8 == 0) && (g.dv_seq_stride % 8 == 0) &&
                         ((((uintptr_t)g.dk | (uintptr_t)g.dv) & 15) == 0);
#pragma unroll
    for (int ps = 0; ps < 32 / RPP; ps++) {
      const int row = ps * RPP + lane / CPR, ch = lane % CPR;
      const int key = wave * 32 + row;
      if (key < Lk && wide_ok) {
        *reinterpret_cast<uint4*>(dkp + (long)key * g.lddk + ch * 8) = *reinterpret_cast<const uint4*>(ek + row * RSE + ch * 8);
        *reinterpret_cast<uint4*>(dvp + (long)key * g.lddv + ch * 8) = *reinterpret_cast<const uint4*>(ev + row * RSE + ch * 8);
      } else if (key < Lk && !ABL(g, 64)) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          dkp[(long)key * g.lddk + ch * 8 + e] = ek[row * RSE + ch * 8 + e];
          dvp[(long)key * g.lddv + ch * 8 + e] = ev[row * RSE + ch * 8 + e];
        }
      }
    }
  } else if (mykey < Lk && !ABL(g, 64)) {
    const long dkofs = (long)seq * g.dk_seq_stride + (long)mykey * g.lddk + head * DH;
    const long dvofs = (long)seq * g.dv_seq_stride + (long)mykey * g.lddv + head * DH;
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int dh0 = n * 32 + 8 * c + 4 * lh;
        hftt_store4(g.dk, dkv_bf, dkofs + dh0, dKT[n][4 * c], dKT[n][4 * c + 1], dKT[n][4 * c + 2], dKT[n][4 * c + 3]);
        hftt_store4(g.dv, dkv_bf, dvofs + dh0, dVT[n][4 * c], dVT[n][4 * c + 1], dVT[n][4 * c + 2], dVT[n][4 * c + 3]);
      }
  }
}

template <int KT, int DH, int NPASS, bool HB>
int launch_ab(const hftt_attn_desc& d, hipStream_t st) {
  using Cfg = AbCfg<KT, DH, NPASS>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<KT, DH, NPASS, HB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) { hftt_set_error("attn_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_bwd_kernel<KT, DH, NPASS, HB>), dim3((unsigned)(d.n_seq * d.n_heads)), dim3(Cfg::NTHR), Cfg::LDS_BYTES, st, d);
  HFTT_CHECK_LAUNCH("attn_bwd");
  return 0;
}

template <int DH, int NPASS, bool HB>
int dispatch_ab(const hftt_attn_desc& d, hipStream_t st) {
  const int kt = (d.Lk + 31) / 32;
  if (kt <= 1) return launch_ab<1, DH, NPASS, HB>(d, st);
  if (kt <= 2) return launch_ab<2, DH, NPASS, HB>(d, st);
  if (kt <= 3) return launch_ab<3, DH, NPASS, HB>(d, st);
  if (kt <= 4) return launch_ab<4, DH, NPASS, HB>(d, st);
  return launch_ab<8, DH, NPASS, HB>(d, st);
}

}  // namespace

extern "C" int hftt_attn_bwd(const hftt_attn_desc* d0, void* stream) {
  int rc = hftt_attn_check(d0, true);
  if (rc) return rc;
  hftt_attn_desc dd = *d0;
  dd.pad = 0;
#ifdef HFTT_ATTN_ABLATE
  if (const char* e = getenv("HFTT_ATTN_ABLATE")) dd.pad = (uint32_t)atoi(e);
#endif
  const hftt_attn_desc* d = &dd;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool hb = hftt_attn_hb_form(*d);      // the forward's predicate: lse[0] is the raw maximum exactly when the forward ran this form
  HFTT_REQUIRE(!(d->io_flags & HFTT_ATTN_DQ_BF16) || (d->lddq % 2 == 0 && d->dq_seq_stride % 2 == 0), "attn_bwd: bf16 dq needs even strides");
  if (d->npass == 2) return hftt_x3_attn_bwd(*d, st);
  if (d->npass == 3) return d->dh == 64 ? dispatch_ab<64, 3, false>(*d, st) : dispatch_ab<32, 3, false>(*d, st);
  if (d->dh == 64) return hb ? dispatch_ab<64, 1, true>(*d, st) : dispatch_ab<64, 1, false>(*d, st);
  return hb ? dispatch_ab<32, 1, true>(*d, st) : dispatch_ab<32, 1, false>(*d, st);
}
