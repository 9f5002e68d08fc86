// Fused multi-head attention forward (gfx950), one workgroup per (sequence, head).
//   P = softmax(Q K^T / sqrt(dh)),  out = dropout(P) V,  lse = logsumexp       -- include/hftt_hip.h (hftt_attn_fwd)
// Sequence lengths on this path are fixed and short (256 bins / 88 notes / 128 frames), so the whole K/V of a
// (sequence, head) is staged once in LDS as bf16 (hi[/lo]) and each wave owns 32 query rows at a time:
//   S^T tile = mfma(A = K rows from LDS, B = Q rows in registers): lane = one query, its keys in the 16 accumulator
//   registers  -> the softmax row reduction is register-local plus ONE cross-half shuffle (lane ^ 32);
//   the probability tile is then fed straight back as the A operand of the PV product (accumulator-as-operand,
//   probe T4), V fragments come from the row-major LDS image through ds_read_b64_tr_b16 (probe T5).
#include <type_traits>
#include "hftt_common.h"
#include "hftt_host.h"
#include "x3_internal.h"
#include "../../include/hftt_hip.h"
#include <math.h>

// Ablation build (tools/ablate_attn.sh compiles this file with -DHFTT_ATTN_ABLATE into its own library): the descriptor's pad field
// (from HFTT_ATTN_ABLATE in the environment) switches single mechanisms off, so that their cost can be read from the timing difference
// (results are then garbage).   1 no K/V staging loads   2 no QK^T MFMAs   4 no softmax arithmetic   8 no dropout   16 no PV MFMAs
// 32 no output / statistics stores
#ifdef HFTT_ATTN_ABLATE
#define ABL(g, bit) (((g).pad & (bit)) != 0)
#else
#define ABL(g, bit) false
#endif

int hftt_attn_fwd8_try(const hftt_attn_desc& d, hipStream_t st);      // attn_fwd8.hip

namespace {

// npass == 1: K/V as bf16 planes (layout notes below).  npass == 3 (parity): K/V stay fp32 in LDS and every product runs on
// v_mfma_f32_32x32x2_f32 (one f32 per lane per operand): K rows of DH+1 floats (conflict-free column reads), V row-major.
template <int KT, int DH, int NPASS>
struct AfCfg {
  static constexpr bool F32 = (NPASS == 3);
  static constexpr int LKP = KT * 32;
  static constexpr int RSK = F32 ? DH + 1 : DH + 8;   // bf16: 144 B / 80 B rows, conflict-free ds_read_b128
  // bf16 tr16 reads are conflict-free when the row stride is 64 (mod 128) bytes
  // (256 keys, dh 64: 144 B rows instead -- 2-way conflicts on some tr reads, but K+V then fit 74 KB = two workgroups per CU)
  static constexpr int RSV = F32 ? DH : ((DH == 64) ? ((KT == 8) ? 72 : 96) : 32);
  static constexpr int K_ELEMS = LKP * RSK;
  static constexpr int V_ELEMS = LKP * RSV;
  static constexpr int LDS_BYTES = (K_ELEMS + V_ELEMS) * (F32 ? 4 : 2);
};

// HB: q, k, v and out are all stored as bf16 -> 16-byte loads straight into LDS / fragments, packed 4-byte output stores
template <int KT, int DH, int NPASS, bool HB>
__global__ __launch_bounds__(256, (NPASS == 3 ? 1 : 2)) void attn_fwd_kernel(const hftt_attn_desc g) {
  using Cfg = AfCfg<KT, DH, NPASS>;
  constexpr bool F32 = Cfg::F32;
  static_assert(!(F32 && HB), "bf16-stored tensors are a bf16-mode feature");
  constexpr int RSK = Cfg::RSK, RSV = Cfg::RSV, LKP = Cfg::LKP;
  constexpr int KS = DH / 16;   // bf16 k-steps of the QK^T product
  constexpr int NT = DH / 32;   // output column tiles
  constexpr int HD = DH / 2;    // fp32 path: dh elements per lane half
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* Ks16 = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Vs16 = Ks16 + Cfg::K_ELEMS;
  float* Ks32 = reinterpret_cast<float*>(smem);
  float* Vs32 = Ks32 + Cfg::K_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int seq = blockIdx.x / g.n_heads, head = blockIdx.x % g.n_heads;
  const int Lq = g.Lq, Lk = g.Lk;
  const bool q_bf = !F32 && (g.io_flags & HFTT_ATTN_Q_BF16), kv_bf = !F32 && (g.io_flags & HFTT_ATTN_KV_BF16), o_bf = !F32 && (g.io_flags & HFTT_ATTN_O_BF16);

  // HB: the Q fragments of a wave's NEXT query block are fetched while the current one is computed (the first block's before the K/V staging):
  // a workgroup lives for two blocks per wave, so a load waited for at its point of use is exposed in full every time
  bf16x8 qn[HB ? DH / 16 : 1];
  auto q_prefetch = [&](int qb) {
    if (HB) {
      const int qr = qb * 32 + lr;
      const int qc = qr < Lq ? qr : Lq - 1;            // clamped: the loads are unconditional (a block past the end re-reads the last row)
      const unsigned short* p = reinterpret_cast<const unsigned short*>(g.q) + (long)seq * g.q_seq_stride + (long)qc * g.ldq + head * DH + 8 * lh;
#pragma unroll
      for (int s = 0; s < DH / 16; s++) qn[s] = *reinterpret_cast<const bf16x8*>(p + 16 * s);
    }
  };
  q_prefetch(wave);
  // ---- stage K and V of this (seq, head) into LDS: 8 + 8 x 16 B loads in flight per thread, then convert + store ----
  {
    const long kofs = (long)seq * g.k_seq_stride + head * DH;      // element offsets (the tensors may be fp32 or bf16)
    const long vofs = (long)seq * g.v_seq_stride + head * DH;
    if (ABL(g, 1)) {
    } else if (HB) {
      constexpr int S8R = DH / 8;                    // 16-byte slots (8 bf16) per key row
      constexpr int TOTAL8 = LKP * S8R;
      constexpr int UB8 = (TOTAL8 / 256) < 8 ? ((TOTAL8 / 256) < 1 ? 1 : (TOTAL8 / 256)) : 8;
      const unsigned short* kp = reinterpret_cast<const unsigned short*>(g.k) + kofs;
      const unsigned short* vp = reinterpret_cast<const unsigned short*>(g.v) + vofs;
      for (int base = 0; base < TOTAL8; base += 256 * UB8) {
        uint4 kf[UB8], vf[UB8];
#pragma unroll
        for (int u = 0; u < UB8; u++) {
          const int i = base + tid + 256 * u;
          const int ic = i < TOTAL8 ? i : TOTAL8 - 1;
          const int key = ic / S8R, c8 = ic % S8R;
          const int kc = key < Lk ? key : Lk - 1;
          kf[u] = *reinterpret_cast<const uint4*>(kp + (long)kc * g.ldk + c8 * 8);
          vf[u] = *reinterpret_cast<const uint4*>(vp + (long)kc * g.ldv + c8 * 8);
        }
#pragma unroll
        for (int u = 0; u < UB8; u++) {
          const int i = base + tid + 256 * u;
          if (i < TOTAL8) {
            const int key = i / S8R, c8 = i % S8R;
            // keys past Lk become zero rows here, not right after the load (a select there makes hipcc wait for every load in turn)
            if (key >= Lk) { kf[u] = make_uint4(0u, 0u, 0u, 0u); vf[u] = kf[u]; }
            *reinterpret_cast<uint4*>(Ks16 + key * RSK + c8 * 8) = kf[u];
            *reinterpret_cast<uint4*>(Vs16 + key * RSV + c8 * 8) = vf[u];
          }
        }
      }
    } else {
    constexpr int F4R = DH / 4;
    constexpr int TOTAL = LKP * F4R;
    constexpr int UB = (TOTAL / 256) < 8 ? (TOTAL / 256) : 8;     // TOTAL is a multiple of 256 for every (KT, DH)
    for (int base = 0; base < TOTAL; base += 256 * UB) {
      float4 kf[UB], vf[UB];
#pragma unroll
      for (int u = 0; u < UB; u++) {
        const int i = base + tid + 256 * u;
        const int key = i / F4R, c4 = i % F4R;
        const int kc = key < Lk ? key : Lk - 1;       // clamped address + select: loads stay unconditional
        kf[u] = hftt_load4(g.k, kv_bf, kofs + (long)kc * g.ldk + c4 * 4);
        vf[u] = hftt_load4(g.v, kv_bf, vofs + (long)kc * g.ldv + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < UB; u++) {
        const int i = base + tid + 256 * u;
        const int key = i / F4R, c4 = i % F4R;
        if (key >= Lk) { kf[u] = make_float4(0.f, 0.f, 0.f, 0.f); vf[u] = kf[u]; }
        if (F32) {
          float* kd = Ks32 + key * RSK + c4 * 4;
          kd[0] = kf[u].x; kd[1] = kf[u].y; kd[2] = kf[u].z; kd[3] = kf[u].w;
          *reinterpret_cast<float4*>(Vs32 + key * RSV + c4 * 4) = vf[u];
        } else {
          uint2 w;
          w.x = f2bf(kf[u].x) | ((unsigned)f2bf(kf[u].y) << 16); w.y = f2bf(kf[u].z) | ((unsigned)f2bf(kf[u].w) << 16);
          *reinterpret_cast<uint2*>(Ks16 + key * RSK + c4 * 4) = w;
          w.x = f2bf(vf[u].x) | ((unsigned)f2bf(vf[u].y) << 16); w.y = f2bf(vf[u].z) | ((unsigned)f2bf(vf[u].w) << 16);
          *reinterpret_cast<uint2*>(Vs16 + key * RSV + c4 * 4) = w;
        }
      }
    }
    }
  }
  __syncthreads();

  const float scale = 1.0f / sqrtf((float)DH);
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const int nqb = (Lq + 31) / 32;
  const bool vec_probs = (Lk % 4) == 0;
  const uint64_t hk = hftt_hash_key(g.drop_seed, g.drop_site);
  const bool quad_ok = (Lk & 3) == 0 && (((uint64_t)g.n_seq * (uint64_t)g.n_heads * (uint64_t)Lq * (uint64_t)Lk) >> 34) == 0;

  for (int qb = wave; qb < nqb; qb += 4) {
    const int qrow = qb * 32 + lr;                     // this lane's query (as the B-operand column)
    const int qrow_c = qrow < Lq ? qrow : Lq - 1;
    const long qofs = (long)seq * g.q_seq_stride + (long)qrow_c * g.ldq + head * DH;
    const float* qp = g.q + qofs;                      // valid as a pointer only when q is fp32
    // ---- S^T = K . Q^T  (rows = keys in registers, column = this lane's query) ----
    f32x16 sacc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) sacc[kt][r] = 0.f;
    if (F32) {
      float qf[HD];                                    // Q[q][HD*lh + t], unscaled
#pragma unroll
      for (int t4 = 0; t4 < HD / 4; t4++) {
        const float4 f = *reinterpret_cast<const float4*>(qp + HD * lh + 4 * t4);
        qf[4 * t4] = f.x; qf[4 * t4 + 1] = f.y; qf[4 * t4 + 2] = f.z; qf[4 * t4 + 3] = f.w;
      }
#pragma unroll
      for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int t = 0; t < HD; t++)
          sacc[kt] = mfma32_f32(Ks32[(kt * 32 + lr) * RSK + HD * lh + t], qf[t], sacc[kt]);
    } else {
      bf16x8 qh[KS];
      if (HB) {
#pragma unroll
        for (int s = 0; s < KS; s++) qh[s] = qn[s];      // unscaled: the scale is folded into the exponent below
        q_prefetch(qb + 4);
      } else
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const float4 f0 = hftt_load4(g.q, q_bf, qofs + 16 * s + 8 * lh);
        const float4 f1 = hftt_load4(g.q, q_bf, qofs + 16 * s + 8 * lh + 4);
        const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
        for (int e = 0; e < 8; e++) qh[s][e] = (short)f2bf(v[e]);
      }
      if (!ABL(g, 2))
#pragma unroll
      for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int s = 0; s < KS; s++)
          sacc[kt] = mfma32(lds_read_b128(Ks16 + (kt * 32 + lr) * RSK + 16 * s + 8 * lh), qh[s], sacc[kt]);
    }
    // ---- softmax over keys (register-local + one cross-half exchange) ----
    // HB: the maximum is taken over the RAW scores (scale > 0 keeps the order), subtracted from them EXACTLY (the maximum element gives 0 and
    // p = 1 whatever the magnitude of the scores: a rounded `s*c2 - m2` is off by ulp(m2), which at |scores| ~ 1e9 -- seen in training --
    // is > 126 and underflows the whole row), and the scale is folded with log2(e) into the multiply in front of v_exp_f32:
    // p = 2^((s - max) * c2), c2 = scale*log2e.  lse[0] holds that RAW maximum in this mode; the backward recomputes bitwise the same P.
    constexpr float LOG2E = 1.4426950408889634f;
    const float c2 = scale * LOG2E;
    if (!HB) {
      // energy = (Q K^T) / sqrt(dh), in the reference's order (model_spec2midi.py:354: the matmul first, then the division): folding the
      // scale into q instead moves every logit by its own rounding (6e-8 relative: 0.006 at the 1e5 of a first-layer row, 0.6 % in p), and the
      // backward -- which recomputes P from the raw product -- would then see other probabilities than the forward used.
      // 1/8 is exact; for sqrt(32) the fp32 mode divides as the reference does.
#pragma unroll
      for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) sacc[kt][r] = hftt_attn_scaled<DH, F32>(sacc[kt][r]);
    }
    // 4*lh as a value the optimiser cannot see through: every per-register key number below is then (compile-time constant + lh4) formed
    // where it is used.  Left visible, LICM hoists all KT*16 of them (and their 64-bit forms) out of the query-block loop and spills them.
    int lh4 = 4 * lh;
    asm volatile("" : "+v"(lh4));
    float mx = -INFINITY;
    if (!ABL(g, 4)) {
    if (Lk < LKP) {                                   // (wave-uniform) key padding of the last tile
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
    }
    float sum = 0.f;
    if (!ABL(g, 4)) {
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = HB ? __builtin_amdgcn_exp2f((sacc[kt][r] - mx) * c2) : (F32 ? expf(sacc[kt][r] - mx) : __expf(sacc[kt][r] - mx));
        sacc[kt][r] = p;
        sum += p;
      }
    sum = xor32_sum(sum);
    }
    const float inv = ABL(g, 4) ? 1.0f : 1.0f / sum;
    if (lh == 0 && qrow < Lq && !ABL(g, 32)) {      // row statistics for the backward recompute: P = exp(s - max) * inv  (bitwise the forward P)
      float* st = g.lse + (((long)seq * g.n_heads + head) * Lq + qrow) * 2;
      st[0] = mx; st[1] = inv;
    }

    const long prow = (((long)seq * g.n_heads + head) * Lq + qrow) * (long)Lk;   // element index base of this query's row
    // Normalisation and dropout in one multiply: kept elements are scaled by inv / (1 - p), dropped ones become zero.
    // Registers 4c..4c+3 of a lane are four CONSECUTIVE keys of its row, starting at a multiple of 4; with Lk % 4 == 0 the row starts at such
    // an element index too, so the four are one hash quad (hftt_keep: byte idx&3 of hash(idx>>2)).  `quad_ok` also asks that no quad index
    // of this launch reaches 2^32 (the high word is then zero and is not mixed); otherwise every element takes the general form.
    // (The branches stay inside the (kt, c) loop: arms that rewrite the whole tile make hipcc keep two copies of it and spill.)
    const float nrm = (g.probs != nullptr) ? inv_keep : inv * inv_keep;
    const uint32_t q0lo = (uint32_t)((uint64_t)prow >> 2) + (uint32_t)(lh4 >> 2);     // quad index of this lane's first key (register 0)
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
        if (g.drop_p > 0.f && !ABL(g, 8)) {        // (wave-uniform)
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
    if (F32) {
      // step r: k=0 <-> key row(r,0), k=1 <-> key row(r,1); each lane feeds its own P^T register as A[i=q][k=lh]
#pragma unroll
      for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = kt * 32 + acc_row32(r, lh);
#pragma unroll
          for (int n = 0; n < NT; n++) oacc[n] = mfma32_f32(sacc[kt][r], Vs32[key * RSV + n * 32 + lr], oacc[n]);
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < KT; kt++) {
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          bf16x8 xh;
#pragma unroll
          for (int e = 0; e < 8; e++) xh[e] = (short)f2bf(sacc[kt][8 * s2 + e]);
          if (!ABL(g, 16))
#pragma unroll
          for (int n = 0; n < NT; n++) {
            const int col = n * 32 + 16 * (gi & 1) + 4 * pp;
            const int r0 = kt * 32 + 16 * s2 + 4 * lh + qq;     // + 8 for the second half of the fragment
            const bf16x8 vh = join4(lds_read_tr16(Vs16 + r0 * RSV + col), lds_read_tr16(Vs16 + (r0 + 8) * RSV + col));
            oacc[n] = mfma32(xh, vh, oacc[n]);
          }
        }
      }
    }
    const long oofs = (long)seq * g.o_seq_stride + head * DH;
    if (ABL(g, 32)) {
    } else if (HB) {
      // lanes 2i / 2i+1 hold adjacent columns: exchange so that each lane stores one packed pair (4 bytes) per register pair
      unsigned short* op = reinterpret_cast<unsigned short*>(g.out) + oofs;
      const bool odd = lane & 1;
      const bool full_qb = qb * 32 + 32 <= Lq;          // wave-uniform: no per-store bounds test for interior query blocks
#pragma unroll
      for (int n = 0; n < NT; n++)
#pragma unroll
        for (int rp = 0; rp < 8; rp++) {
          const unsigned pk = pair_rows_to_cols(oacc[n][2 * rp], oacc[n][2 * rp + 1], odd);
          const int q = qb * 32 + acc_row32(2 * rp + (odd ? 1 : 0), lh);
          if (full_qb || q < Lq) *reinterpret_cast<unsigned*>(op + (long)q * g.ldo + n * 32 + (lr & ~1)) = pk;
        }
    } else {
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int q = qb * 32 + acc_row32(r, lh);
        if (q < Lq) hftt_store1(g.out, o_bf, oofs + (long)q * g.ldo + n * 32 + lr, oacc[n][r]);
      }
    }
  }
}

template <int KT, int DH, int NPASS, bool HB>
int launch_af(const hftt_attn_desc& d, hipStream_t st) {
  using Cfg = AfCfg<KT, DH, NPASS>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<KT, DH, NPASS, HB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) { hftt_set_error("attn_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_fwd_kernel<KT, DH, NPASS, HB>), dim3((unsigned)(d.n_seq * d.n_heads)), dim3(256), Cfg::LDS_BYTES, st, d);
  HFTT_CHECK_LAUNCH("attn_fwd");
  return 0;
}

template <int DH, int NPASS, bool HB>
int dispatch_af(const hftt_attn_desc& d, hipStream_t st) {
  const int kt = (d.Lk + 31) / 32;
  if (kt <= 1) return launch_af<1, DH, NPASS, HB>(d, st);
  if (kt <= 2) return launch_af<2, DH, NPASS, HB>(d, st);
  if (kt <= 3) return launch_af<3, DH, NPASS, HB>(d, st);
  if (kt <= 4) return launch_af<4, DH, NPASS, HB>(d, st);
  return launch_af<8, DH, NPASS, HB>(d, st);
}

}  // namespace

int hftt_attn_check(const hftt_attn_desc* d, bool bwd) {
  HFTT_REQUIRE(d != nullptr, "attn: null descriptor");
  HFTT_REQUIRE(d->n_seq > 0 && d->n_heads > 0, "attn: bad batch");
  HFTT_REQUIRE(d->Lq >= 1 && d->Lq <= 256 && d->Lk >= 1 && d->Lk <= 256, "attn: Lq=%d Lk=%d must be in 1..256", d->Lq, d->Lk);
  HFTT_REQUIRE(d->dh == 32 || d->dh == 64, "attn: head_dim=%d must be 32 or 64", d->dh);
  HFTT_REQUIRE(d->npass == 1 || d->npass == 2 || d->npass == 3, "attn: npass must be 1 (bf16), 2 (split fp16 / bf16) or 3 (fp32)");
  HFTT_REQUIRE(d->q && d->k && d->v && d->out && d->lse, "attn: null operand");
  HFTT_REQUIRE(d->ldq % 4 == 0 && d->ldk % 4 == 0 && d->ldv % 4 == 0 && d->ldo % 4 == 0, "attn: row strides must be multiples of 4");
  HFTT_REQUIRE(d->q_seq_stride % 4 == 0 && d->k_seq_stride % 4 == 0 && d->v_seq_stride % 4 == 0, "attn: seq strides must be multiples of 4");
  HFTT_REQUIRE((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v | (uintptr_t)d->out) & 15) == 0, "attn: operands must be 16-byte aligned");
  HFTT_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "attn: drop_p out of range");
  const unsigned bf_flags = HFTT_ATTN_Q_BF16 | HFTT_ATTN_KV_BF16 | HFTT_ATTN_O_BF16 | HFTT_ATTN_DQ_BF16 | HFTT_ATTN_DKV_BF16;
  const unsigned pl_flags = HFTT_ATTN_Q_F16PAIR | HFTT_ATTN_KV_F16PAIR;
  HFTT_REQUIRE((d->io_flags & bf_flags) == 0 || d->npass == 1, "attn: bf16-stored tensors need npass == 1");
  HFTT_REQUIRE((d->io_flags & pl_flags) == 0 || d->npass == 2, "attn: f16-pair planes need npass == 2");
  HFTT_REQUIRE((d->io_flags & ~(bf_flags | pl_flags)) == 0, "attn: unknown io_flags 0x%x", d->io_flags);
  if (bwd) {
    HFTT_REQUIRE(d->dout && d->dq && d->dk && d->dv, "attn_bwd: null gradient operand");
    HFTT_REQUIRE(d->lddq % 4 == 0 && d->lddk % 4 == 0 && d->lddv % 4 == 0, "attn_bwd: row strides must be multiples of 4");
    HFTT_REQUIRE((((uintptr_t)d->dout | (uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & 15) == 0, "attn_bwd: operands must be 16-byte aligned");
  }
  return 0;
}

extern "C" int hftt_attn_fwd(const hftt_attn_desc* d0, void* stream) {
  int rc = hftt_attn_check(d0, false);
  if (rc) return rc;
  hftt_attn_desc dd = *d0;
  dd.pad = 0;
#ifdef HFTT_ATTN_ABLATE
  if (const char* e = getenv("HFTT_ATTN_ABLATE")) dd.pad = (uint32_t)atoi(e);
#endif
  const hftt_attn_desc* d = &dd;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->npass == 2) return hftt_x3_attn_fwd(*d, st);
#ifndef HFTT_ATTN_ABLATE
  {                                                   // the long-row form (attn_fwd8.hip) where it applies
    const int rc8 = hftt_attn_fwd8_try(*d, st);
    if (rc8 >= 0) return rc8;
  }
#endif
  const bool hb = hftt_attn_hb_form(*d);
  if (d->npass == 3) return d->dh == 64 ? dispatch_af<64, 3, false>(*d, st) : dispatch_af<32, 3, false>(*d, st);
  if (d->dh == 64) return hb ? dispatch_af<64, 1, true>(*d, st) : dispatch_af<64, 1, false>(*d, st);
  return hb ? dispatch_af<32, 1, true>(*d, st) : dispatch_af<32, 1, false>(*d, st);
}

bool hftt_attn_hb_form(const hftt_attn_desc& d) {
  const unsigned all_half = HFTT_ATTN_Q_BF16 | HFTT_ATTN_KV_BF16 | HFTT_ATTN_O_BF16;
  return (d.io_flags & all_half) == all_half && d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 8 == 0 &&
         d.q_seq_stride % 8 == 0 && d.k_seq_stride % 8 == 0 && d.v_seq_stride % 8 == 0 && d.o_seq_stride % 8 == 0;
}
