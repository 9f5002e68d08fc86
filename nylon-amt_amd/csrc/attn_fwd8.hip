// Fused multi-head attention forward, long-row form: 129 .. 256 keys, head dim 64, all tensors bf16 (the encoder's 256 x 256 self-attention).
// Same contract and arithmetic as attn_fwd.hip (which stays for every other shape, the attention-map output and the parity mode).
//
// Why.  In attn_fwd.hip a wave holds the whole score row of its 32 queries (KT * 16 = 128 accumulator registers at 256 keys), so a
// workgroup is 4 waves x 2 query blocks and two workgroups per CU leave TWO waves per SIMD: staging, QK^T, the softmax arithmetic, PV and
// the stores of a workgroup run one after the other (ablation, tools/ablate_attn.sh: the parts add up to the whole) and one partner wave
// cannot cover them.  Here the row is walked tile by tile (32 keys = 16 accumulator registers), twice:
//   pass A   S tile = K . Q^T -> row maximum only;
//   pass B   S tile again -> p = 2^((s - max)*c2) (the FINAL maximum, so no rescaling) -> row sum, dropout -> P tile . V accumulated.
// 32 more MFMAs per query block (QK^T twice), but a wave fits 128 registers: 8 waves per workgroup, one query block each, two
// workgroups per CU = FOUR waves per SIMD.  The row normalisation 1/sum is known only at the end and lives in the lane of its QUERY,
// while the output accumulators hold a query per REGISTER row: it crosses through 128 bytes of LDS per wave.
#include <stdlib.h>
#include <type_traits>
#include "hftt_common.h"
#include "hftt_host.h"
#include "x3_internal.h"
#include "../../include/hftt_hip.h"
#include <math.h>

namespace {

constexpr int DH = 64, LKP = 256;
constexpr int RSK = DH + 8;          // 144-byte rows: conflict-free ds_read_b128
constexpr int RSV = 72;              // (see attn_fwd.hip: K + V = 74 KB, two workgroups per CU)
constexpr int K_ELEMS = LKP * RSK, V_ELEMS = LKP * RSV;
constexpr int LDS_BYTES8 = (K_ELEMS + V_ELEMS) * 2 + 8 * 32 * 4;

__global__ __launch_bounds__(512, 2) void attn_fwd8_kernel(const hftt_attn_desc g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* Ks16 = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Vs16 = Ks16 + K_ELEMS;
  float* inv_s = reinterpret_cast<float*>(smem + (size_t)(K_ELEMS + V_ELEMS) * 2);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int gi = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int seq = blockIdx.x / g.n_heads, head = blockIdx.x % g.n_heads;
  const int Lq = g.Lq, Lk = g.Lk;
  const int qb = wave;                                 // one 32-query block per wave (Lq <= 256)
  const bool active = qb * 32 < Lq;                    // (wave-uniform)

  // this wave's Q fragments: in flight while K / V are staged
  const int qrow = qb * 32 + lr;
  const int qrow_c = qrow < Lq ? qrow : Lq - 1;
  bf16x8 qh[4];
  {
    const unsigned short* p = reinterpret_cast<const unsigned short*>(g.q) + (long)seq * g.q_seq_stride + (long)qrow_c * g.ldq + head * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; s++) qh[s] = *reinterpret_cast<const bf16x8*>(p + 16 * s);
  }
  // ---- stage K and V of this (seq, head): 4 + 4 x 16 B per thread in flight, then store (rows past Lk become zeros)
  {
    const unsigned short* kp = reinterpret_cast<const unsigned short*>(g.k) + (long)seq * g.k_seq_stride + head * DH;
    const unsigned short* vp = reinterpret_cast<const unsigned short*>(g.v) + (long)seq * g.v_seq_stride + head * DH;
    uint4 kf[4], vf[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = tid + 512 * u;
      const int key = i >> 3, c8 = i & 7;
      const int kc = key < Lk ? key : Lk - 1;
      kf[u] = *reinterpret_cast<const uint4*>(kp + (long)kc * g.ldk + c8 * 8);
      vf[u] = *reinterpret_cast<const uint4*>(vp + (long)kc * g.ldv + c8 * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = tid + 512 * u;
      const int key = i >> 3, c8 = i & 7;
      if (key >= Lk) { kf[u] = make_uint4(0u, 0u, 0u, 0u); vf[u] = kf[u]; }
      *reinterpret_cast<uint4*>(Ks16 + key * RSK + c8 * 8) = kf[u];
      *reinterpret_cast<uint4*>(Vs16 + key * RSV + c8 * 8) = vf[u];
    }
  }
  __syncthreads();
  if (!active) return;                                 // (no barrier follows)

  const float scale = 1.0f / sqrtf((float)DH);
  constexpr float LOG2E = 1.4426950408889634f;
  const float c2 = scale * LOG2E;
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const uint64_t hk = hftt_hash_key(g.drop_seed, g.drop_site);
  const bool quad_ok = (Lk & 3) == 0 && (((uint64_t)g.n_seq * (uint64_t)g.n_heads * (uint64_t)Lq * (uint64_t)Lk) >> 34) == 0;
  int lh4 = 4 * lh;                                    // opaque (attn_fwd.hip: keeps per-register key numbers out of LICM's reach)
  asm volatile("" : "+v"(lh4));
  const int lkm = Lk - lh4;

  // S^T tile (keys 32*kt .. 32*kt+31 in the registers, this lane's query the column), raw scores
  auto scores = [&](int kt, f32x16& s) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
      s = mfma32(lds_read_b128(Ks16 + (kt * 32 + lr) * RSK + 16 * ks + 8 * lh), qh[ks], s);
    if (Lk < LKP) {                                    // (wave-uniform) key padding
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (kt * 32 + (r & 3) + 8 * (r >> 2) >= lkm) s[r] = -INFINITY;
    }
  };

  // ---- pass A: the row maximum
  float mx = -INFINITY;
#pragma unroll 1
  for (int kt = 0; kt < 8; kt++) {
    f32x16 s;
    scores(kt, s);
#pragma unroll
    for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[r]);
  }
  mx = xor32_max(mx);                                  // the RAW maximum (what lse[0] holds in the bf16 / x3 kernels: attn_fwd.hip)

  // ---- pass B: probabilities, row sum, dropout, P . V
  const long prow = (((long)seq * g.n_heads + head) * Lq + qrow) * (long)Lk;
  const uint32_t q0lo = (uint32_t)((uint64_t)prow >> 2) + (uint32_t)(lh4 >> 2);
  float sum = 0.f;
  f32x16 oacc[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[n][r] = 0.f;
#pragma unroll 1
  for (int kt = 0; kt < 8; kt++) {
    f32x16 s;
    scores(kt, s);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float p = __builtin_amdgcn_exp2f((s[r] - mx) * c2);
      s[r] = p;
      sum += p;
    }
    if (g.drop_p > 0.f) {                              // (wave-uniform) unnormalised: the row's 1/sum is applied to the output rows at the end
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (quad_ok) {
          const uint32_t w = hftt_hash_mix(hk, q0lo + (uint32_t)(kt * 8 + 2 * c), 0u);
          const float a0 = s[4 * c], a1 = s[4 * c + 1], a2 = s[4 * c + 2], a3 = s[4 * c + 3];
          s[4 * c] = (w & 0xFFu) < thr ? a0 * inv_keep : 0.f;
          s[4 * c + 1] = ((w >> 8) & 0xFFu) < thr ? a1 * inv_keep : 0.f;
          s[4 * c + 2] = ((w >> 16) & 0xFFu) < thr ? a2 * inv_keep : 0.f;
          s[4 * c + 3] = (w >> 24) < thr ? a3 * inv_keep : 0.f;
        } else {
          const int key0 = kt * 32 + 8 * c + lh4;
#pragma unroll
          for (int e = 0; e < 4; e++)
            s[4 * c + e] = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)(prow + key0 + e), thr) ? s[4 * c + e] * inv_keep : 0.f;
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      bf16x8 xh;
#pragma unroll
      for (int e = 0; e < 8; e++) xh[e] = (short)f2bf(s[8 * s2 + e]);
#pragma unroll
      for (int n = 0; n < 2; n++) {
        const int col = n * 32 + 16 * (gi & 1) + 4 * pp;
        const int r0 = kt * 32 + 16 * s2 + 4 * lh + qq;
        const bf16x8 vh = join4(lds_read_tr16(Vs16 + r0 * RSV + col), lds_read_tr16(Vs16 + (r0 + 8) * RSV + col));
        oacc[n] = mfma32(xh, vh, oacc[n]);
      }
    }
  }
  sum = xor32_sum(sum);
  const float inv = 1.0f / sum;
  if (lh == 0) {
    if (qrow < Lq) {
      float* st = g.lse + (((long)seq * g.n_heads + head) * Lq + qrow) * 2;
      st[0] = mx; st[1] = inv;
    }
    inv_s[wave * 32 + lr] = inv;                       // query lr of this wave's block -> the register rows of the output tile
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float rinv[16];
#pragma unroll
  for (int j4 = 0; j4 < 4; j4++) {
    const float4 t4 = *reinterpret_cast<const float4*>(inv_s + wave * 32 + 8 * j4 + 4 * lh);       // rows 8j + 4lh + {0..3} = registers 4j..4j+3
    rinv[4 * j4] = t4.x; rinv[4 * j4 + 1] = t4.y; rinv[4 * j4 + 2] = t4.z; rinv[4 * j4 + 3] = t4.w;
  }
  // ---- output: lanes 2i / 2i+1 hold adjacent columns: one packed pair (4 bytes) per register pair and lane
  unsigned short* op = reinterpret_cast<unsigned short*>(g.out) + (long)seq * g.o_seq_stride + head * DH;
  const bool odd = lane & 1;
  const bool full_qb = qb * 32 + 32 <= Lq;
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int rp = 0; rp < 8; rp++) {
      const unsigned pk = pair_rows_to_cols(oacc[n][2 * rp] * rinv[2 * rp], oacc[n][2 * rp + 1] * rinv[2 * rp + 1], odd);
      const int q = qb * 32 + acc_row32(2 * rp + (odd ? 1 : 0), lh);
      if (full_qb || q < Lq) *reinterpret_cast<unsigned*>(op + (long)q * g.ldo + n * 32 + (lr & ~1)) = pk;
    }
}

}  // namespace

// -1: not this kernel's shape (attn_fwd.hip handles it); otherwise the launch status.  Called by hftt_attn_fwd after validation.
int hftt_attn_fwd8_try(const hftt_attn_desc& d, hipStream_t st) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("HFTT_ATTN_FWD8"); enabled = !(e && e[0] == '0'); }
  if (!enabled || d.npass != 1 || d.dh != 64 || d.probs != nullptr || !hftt_attn_hb_form(d)) return -1;
  if (d.Lk <= 128 || d.Lk > 256 || d.Lq <= 128 || d.Lq > 256) return -1;                 // long rows, and enough query blocks for the 8 waves
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES8);
    if (e != hipSuccess) { hftt_set_error("attn_fwd8: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_fwd8_kernel, dim3((unsigned)(d.n_seq * d.n_heads)), dim3(512), LDS_BYTES8, st, d);
  HFTT_CHECK_LAUNCH("attn_fwd8");
  return 0;
}
