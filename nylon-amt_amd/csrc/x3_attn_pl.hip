// Attention forward of the split-operand ("x3") mode on PRE-SPLIT operands (gfx950): q, k, v arrive as fp16 hi / lo planes written once by
// the projection that produced them (hftt_strip_linear with HFTT_SL_C_F16PAIR, or hftt_x3_to_planes), not as fp32 that every consumer
// splits again.  Contract: include/hftt_hip.h (hftt_attn_fwd, npass 2, HFTT_ATTN_Q_F16PAIR | HFTT_ATTN_KV_F16PAIR, dh == 64).
//
// Plane layout ("f16 pair"): the 32 fp32 slots (128 bytes) of every aligned 32-column group of a row hold the group's 32 fp16 hi halves
// (64 B) followed by its 32 fp16 lo halves (64 B); every stride of the descriptor keeps its fp32 meaning.  x = hi + lo to 2^-22
// (x3_common.h).  The group is the strip kernels' output tile: a projection's epilogue still writes one whole 128-byte segment per row and
// tile, exactly as for fp32 results (a per-head layout, 64 hi then 64 lo, made every store instruction touch 16 half lines instead of 8
// lines and cost the QKV projection 20 %); the consumers fetch 16-byte pieces at computed addresses anyway.
//
// What this buys (x3_attn.hip staged K / V through the vector ALU: fp32 loads, split, ds_write; 453 us per encoder launch with the
// load phase and the compute phase of a workgroup strictly one after the other, one workgroup per CU):
//   * K and V reach LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU, no ds_write) as 1-KiB pieces of 8 rows x 128 B.  The
//     image has unpadded 128-byte rows; bank conflicts are avoided by an XOR swizzle of the 16-byte chunk index (pl_swz) that is applied on
//     the GLOBAL side of the DMA (lane L of a piece fetches the chunk that belongs at LDS position L), conflict-free for the K rows'
//     ds_read_b128 and for the V rows' ds_read_b64_tr_b16 alike;
//   * the kernel is persistent (one workgroup per CU at 256 keys, two below) and the next item's K is fetched while this item's softmax
//     and P.V run, the next item's V while the next S = K.Q^T runs: three barriers per (sequence, head), memory always in flight;
//   * Q fragments are 16-byte loads of the planes, prefetched one item ahead.
// Arithmetic, summation order, softmax, dropout indexing and outputs are those of x3_attn_fwd_kernel: results are bit-identical.
#include <type_traits>
#include "hftt_common.h"
#include "x3_common.h"
#include "hftt_host.h"
#include "x3_internal.h"
#include "../../include/hftt_hip.h"
#include <math.h>

// Timing switches (results garbage) exist in the ablation build only (-DHFTT_X3_ATTN_ABLATE, tools/ablate_x3p_attn.sh; HFTT_X3P_DEBUG bits):
// 1 no S = K.Q^T MFMAs, 2 no softmax / dropout / P.V phase, 4 no K / V DMA after the first item, 8 no output stores, 16 no Q prefetch loads
#ifdef HFTT_X3_ATTN_ABLATE
#include <stdlib.h>
#define PABL(g, bit) (((g).pad & (bit)) != 0)
#else
#define PABL(g, bit) false
#endif

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// swizzle of the 16-byte chunk index (0..7) of a 128-byte row: bits 1, 2, 3 of the row number, reversed.
//   ds_read_b128 (K rows: lanes = 32 consecutive rows, one chunk column): its 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} hold
//     8 even and 8 odd rows whose (row >> 1) & 7 are all different -> 16 different bank quads;
//   ds_read_b64_tr_b16 (V rows: 32 lanes = 4 consecutive rows x 64 bytes): row parity picks the bank half, bit 1 of the row (the top bit
//     of the swizzle) the 64-byte half of the row -> 4 x 64 B on 64 different banks.
__device__ __forceinline__ int pl_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

// LDS image of one tensor (K or V) of a (sequence, head): groups of FOUR rows, each group 1 KiB = [4 rows x 128 B of hi halves | 4 rows x 128 B
// of lo halves].  One DMA piece fills one group, and its 64 lanes fetch whole 128-byte lines: the 16 lanes of a row take the row's two
// 32-column groups, each 64 B of hi + 64 B of lo (hi-plane-only pieces fetched half lines and ran 10 % slower).
__device__ __forceinline__ int pl_row_off(int row) { return (row >> 2) * 1024 + (row & 3) * 128; }
constexpr int PL_LO = 512;                                    // lo halves of a row: 512 bytes behind its hi halves

template <int KT>
struct PfCfg {
  static constexpr int LKP = KT * 32;
  static constexpr int IMG = LKP * 256;                       // bytes of one tensor's image (hi + lo)
  static constexpr int PIECES = 8 * KT;                       // 1-KiB DMA pieces per tensor
  static constexpr int LDS_BYTES = 2 * IMG + 8 * 32 * 4;      // K | V | one 1/sum per query row and wave
};

// one LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + 16 * lane (M0 carries the LDS base)
__device__ __forceinline__ void pl_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// one tensor (K or V) of one (sequence, head): PIECES pieces dealt over the NW waves (PER per wave, a static count: the waits below count
// them).  base: row 0 of the sequence, this head's first 32-column group; rows past Lk re-fetch row Lk - 1 (finite values; their scores are
// masked to -inf, their probabilities are exactly 0).
template <int KT, int NW>
__device__ __forceinline__ void pl_dma_tensor(const unsigned char* base, long ld_bytes, int Lk, unsigned lds_base, int wave, int lane) {
  constexpr int PIECES = PfCfg<KT>::PIECES;
  constexpr int PER = PIECES / NW;
  static_assert(PIECES % NW == 0, "pieces must divide over the waves");
  // (opaque: left visible, LICM hoists every piece's LDS address and piece number out of the item loop and parks them in scalar
  // registers for the whole kernel, and the per-piece lane offsets in 2 x PER vector registers -- both files spilled)
  asm volatile("" : "+s"(wave), "+v"(lane));
  const int plane = lane >> 5, r4 = (lane >> 3) & 3;
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int pc = wave + NW * u;                               // wave-uniform: rows 4*pc .. 4*pc + 3
    const int row = 4 * pc + r4;
    const int rowc = row < Lk ? row : Lk - 1;
    const int c = (lane & 7) ^ pl_swz(row);                     // 16-byte chunk of the head's 64 halves: group c >> 2, piece c & 3
    pl_glds16(base + (long)rowc * ld_bytes + (c >> 2) * 128 + plane * 64 + (c & 3) * 16, lds_base + (unsigned)(pc * 1024));
  }
}

// NW waves, one 32-query block per wave (host: ceil(Lq / 32) <= NW).  MAP: the attention map (g.probs) is an output.
// DM: the dropout form (0 none, 1 one hash per aligned key quad, 2 per element -- Lk % 4 != 0 or more than 2^34 map elements), chosen by
// the host: as a run-time test inside the tile loop the forms met in phi copies of whole score tiles and a branch per quad
template <int KT, int NW, bool MAP, int DM>
__global__ __launch_bounds__(NW * 64, (KT <= 3) ? 3 : (KT <= 4) ? 2 : 1) void x3p_attn_fwd_kernel(const hftt_attn_desc g, const int n_items) {
  using Cfg = PfCfg<KT>;
  constexpr int E = X3_F16, DH = 64, KS = 4, NT = 2;
  constexpr int IMG = Cfg::IMG, LKP = Cfg::LKP;
  constexpr int PER = Cfg::PIECES / NW;                        // DMA pieces per wave and tensor
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  const unsigned char* Kimg = smem;
  const unsigned char* Vimg = smem + IMG;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Lq = g.Lq, Lk = g.Lk, H = g.n_heads;
#ifdef HFTT_PRIO_SKEW
  if (NW >= 8 && wave < NW / 2) __builtin_amdgcn_s_setprio(HFTT_PRIO_SKEW);
#endif
  const int nqb = (Lq + 31) / 32;
  const bool active = wave < nqb;                              // (wave-uniform) this wave owns query block `wave`
  const int qb = wave;


  const float scale = 1.0f / sqrtf((float)DH);
  const float c2 = scale * LOG2E;
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const bool vec_probs = (Lk % 4) == 0;
  const uint64_t hk = hftt_hash_key(g.drop_seed, g.drop_site);
  const long ldk_b = g.ldk * 4, ldv_b = g.ldv * 4;

  auto k_base = [&](int item) { const int seq = item / H, head = item - seq * H; return reinterpret_cast<const unsigned char*>(g.k + (long)seq * g.k_seq_stride + head * DH); };
  auto v_base = [&](int item) { const int seq = item / H, head = item - seq * H; return reinterpret_cast<const unsigned char*>(g.v + (long)seq * g.v_seq_stride + head * DH); };
  auto q_load = [&](int item, bf16x8 (&qh)[KS], bf16x8 (&ql)[KS]) {
    const int seq = item / H, head = item - seq * H;
    int ln = lane;
    asm volatile("" : "+v"(ln));                               // (the row offset is formed here, not kept in registers across the item loop)
    const int qr = qb * 32 + (ln & 31);
    const int qrc = qr < Lq ? qr : Lq - 1;
    const unsigned char* qp = reinterpret_cast<const unsigned char*>(g.q + (long)seq * g.q_seq_stride + (long)qrc * g.ldq + head * DH) + 16 * (ln >> 5);
#pragma unroll
    for (int s = 0; s < KS; s++) {                             // elements 16s + 8lh ..: group s >> 1, byte 32 * (s & 1) + 16 * lh of its hi half
      qh[s] = *reinterpret_cast<const bf16x8*>(qp + 128 * (s >> 1) + 32 * (s & 1));
      ql[s] = *reinterpret_cast<const bf16x8*>(qp + 128 * (s >> 1) + 32 * (s & 1) + 64);
    }
  };

  // ---- prologue: the first item's K, V and Q ----
  int item = blockIdx.x;
  bf16x8 qh[KS], ql[KS];
  pl_dma_tensor<KT, NW>(k_base(item), ldk_b, Lk, lds0, wave, lane);
  pl_dma_tensor<KT, NW>(v_base(item), ldv_b, Lk, lds0 + IMG, wave, lane);
  q_load(item, qh, ql);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (; item < n_items; item += gridDim.x) {
    const int nxt = (item + (int)gridDim.x < n_items) ? item + (int)gridDim.x : item;      // past the last item: a harmless re-fetch (static counts)
    const int seq = item / H, head = item - seq * H;
    // per-lane byte offsets into the swizzled images, formed per item right where a phase needs them (from values the optimiser cannot see
    // through: hoisted out of the item loop they stay live across both phases and spill at 256 keys)
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    int koff[KS];                                              // K rows kt*32 + lr, chunk 2s + lh
    {
      const int lr_ = lane_s & 31, lh_ = lane_s >> 5;
#pragma unroll
      for (int s = 0; s < KS; s++) koff[s] = pl_row_off(lr_) + (((2 * s + lh_) ^ pl_swz(lr_)) << 4);
    }
    // ---- S^T = K . Q^T  (rows = keys in registers, column = this lane's query) ----
    f32x16 sacc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) sacc[kt][r] = 0.f;
    if (active && !PABL(g, 1)) {
#pragma unroll
      for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int s = 0; s < KS; s++) {
          const unsigned short* p = reinterpret_cast<const unsigned short*>(Kimg + kt * 8192 + koff[s]);
          sacc[kt] = x3_mma<E>(lds_read_b128(p), lds_read_b128(p + PL_LO / 2), qh[s], ql[s], sacc[kt]);
        }
    }
    __syncthreads();                                           // every wave is done with the K image
    if (!PABL(g, 4)) pl_dma_tensor<KT, NW>(k_base(nxt), ldk_b, Lk, lds0, wave, lane);

    // ---- softmax over keys: p = 2^((s - max) * c2); the RAW maximum is subtracted exactly and stored (x3_attn.hip) ----
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    const int lh4 = 4 * (lane_v >> 5);
    const int qrow = qb * 32 + (lane_v & 31);                  // this lane's query (as the B-operand column)
    float mx = -INFINITY;
    if (active) {
      if (Lk < LKP) {
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
    // V of this item has landed once everything but this wave's PER youngest operations (the K pieces just issued) is complete
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    __syncthreads();
    // next item's Q fragments: requested half way through the key tiles (the first half of the score registers is dead by then: requested
    // at the top they cost 32 live registers beside all of sacc), in flight under the rest of this item's P.V
    bf16x8 qnh[KS], qnl[KS];
    constexpr int QAT = (KT == 8) ? 6 : KT / 2;                // the key tile in front of which the request is issued
    if (!active) q_load(nxt, qnh, qnl);

    int voff[NT][2];                                           // V rows 16*s2 + 4*lh + qq (+ 8), elements n*32 + 16*(gi&1) + 4*pp .. + 3
    {
      const int lh_ = lane_v >> 5, gi_ = lane_v >> 4, qq_ = (lane_v & 15) >> 2, pp_ = lane_v & 3;
#pragma unroll
      for (int n = 0; n < NT; n++)
#pragma unroll
        for (int sec = 0; sec < 2; sec++) {
          const int row = 4 * lh_ + qq_ + 8 * sec;
          voff[n][sec] = pl_row_off(row) + (((4 * n + 2 * (gi_ & 1) + (pp_ >> 1)) ^ pl_swz(row)) << 4) + (pp_ & 1) * 8;
        }
    }
    const long sh = (long)seq * H + head;
    const long prow = (sh * Lq + qrow) * (long)Lk;             // element index base of this query's row
    const uint32_t q0lo = (uint32_t)((uint64_t)prow >> 2) + (uint32_t)(lh4 >> 2);
    f32x16 oacc[NT];
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) oacc[n][r] = 0.f;
    float inv = 0.f;
    if (active && !PABL(g, 2)) {
      if (!MAP) {
        // ONE loop over the key tiles: exponentials, row sum, dropout, fp16 split, then the tile's P.V MFMAs with the UNNORMALISED, UNSCALED
        // probabilities; inv_keep / sum multiplies the output rows at the end (x3_attn.hip)
        float sum = 0.f;
        {
#pragma unroll
          for (int kt = 0; kt < KT; kt++) {
            if (kt == QAT && !PABL(g, 16)) q_load(nxt, qnh, qnl);
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
              x3_split8_nc<E>(pt + 8 * s2, ph, pl);                // probabilities: 0 .. 1, no saturation step
#pragma unroll
              for (int n = 0; n < NT; n++) {
                const unsigned char* vb = Vimg + (kt * 32 + 16 * s2) * 256;
                const unsigned short* p0 = reinterpret_cast<const unsigned short*>(vb + voff[n][0]);
                const unsigned short* p1 = reinterpret_cast<const unsigned short*>(vb + voff[n][1]);
                const bf16x8 vh = join4(lds_read_tr16(p0), lds_read_tr16(p1));
                const bf16x8 vl = join4(lds_read_tr16(p0 + PL_LO / 2), lds_read_tr16(p1 + PL_LO / 2));
                oacc[n] = x3_mma<E>(ph, pl, vh, vl, oacc[n]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        sum = xor32_sum(sum);
        inv = 1.0f / sum;
      } else {
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
        inv = 1.0f / sum;
        const float nrm = inv_keep;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int key0 = kt * 32 + 8 * c + lh4;
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
            if (DM == 1) {
              const uint32_t w = hftt_hash_mix(hk, q0lo + (uint32_t)(kt * 8 + 2 * c), 0u);
              const float a0 = sacc[kt][4 * c], a1 = sacc[kt][4 * c + 1], a2 = sacc[kt][4 * c + 2], a3 = sacc[kt][4 * c + 3];
              sacc[kt][4 * c] = (w & 0xFFu) < thr ? a0 * nrm : 0.f;
              sacc[kt][4 * c + 1] = ((w >> 8) & 0xFFu) < thr ? a1 * nrm : 0.f;
              sacc[kt][4 * c + 2] = ((w >> 16) & 0xFFu) < thr ? a2 * nrm : 0.f;
              sacc[kt][4 * c + 3] = (w >> 24) < thr ? a3 * nrm : 0.f;
            } else if (DM == 2) {
#pragma unroll
              for (int e = 0; e < 4; e++)
                sacc[kt][4 * c + e] = hftt_keep(g.drop_seed, g.drop_site, (uint64_t)(prow + key0 + e), thr) ? sacc[kt][4 * c + e] * nrm : 0.f;
            }
          }
        }
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
          if (kt == QAT && !PABL(g, 16)) q_load(nxt, qnh, qnl);
#pragma unroll
          for (int s2 = 0; s2 < 2; s2++) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) pv[e] = sacc[kt][8 * s2 + e];
            bf16x8 ph, pl;
            x3_split8<E>(pv, ph, pl);
#pragma unroll
            for (int n = 0; n < NT; n++) {
              const unsigned char* vb = Vimg + (kt * 32 + 16 * s2) * 256;
              const unsigned short* p0 = reinterpret_cast<const unsigned short*>(vb + voff[n][0]);
              const unsigned short* p1 = reinterpret_cast<const unsigned short*>(vb + voff[n][1]);
              const bf16x8 vh = join4(lds_read_tr16(p0), lds_read_tr16(p1));
              const bf16x8 vl = join4(lds_read_tr16(p0 + PL_LO / 2), lds_read_tr16(p1 + PL_LO / 2));
              oacc[n] = x3_mma<E>(ph, pl, vh, vl, oacc[n]);
            }
          }
        }
      }
    }
    // the next item's Q (and, older than it, the next item's K pieces) must have landed before this wave passes the barrier that lets
    // the next S phase start; the V image is free once every wave is here
#pragma unroll
    for (int s = 0; s < KS; s++) asm volatile("" : "+v"(qnh[s]), "+v"(qnl[s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!PABL(g, 4)) pl_dma_tensor<KT, NW>(v_base(nxt), ldv_b, Lk, lds0 + IMG, wave, lane);

    // ---- epilogue of this item: row statistics and the output rows ----
    if (active && !PABL(g, 8)) {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int lr = lane_e & 31, lh = lane_e >> 5, lh4 = 4 * lh;
      const int qrow = qb * 32 + lr;
      if (lh == 0 && qrow < Lq) {
        float* st = g.lse + (sh * Lq + qrow) * 2;
        st[0] = mx; st[1] = inv;
      }
      const long oofs = (long)seq * g.o_seq_stride + head * DH;
      if (!MAP) {
        // the output tile has the QUERY on its register index (row acc_row32(r, lh)): 1/sum of those rows comes through 128 bytes of LDS
        float* invs = reinterpret_cast<float*>(smem + 2 * IMG) + wave * 32;
        if (lh == 0) invs[lr] = inv * inv_keep;                   // (the kept probabilities entered P.V unscaled)
#pragma unroll
        for (int j4 = 0; j4 < 4; j4++) {
          const float4 i4 = *reinterpret_cast<const float4*>(invs + 8 * j4 + lh4);
          const float iv[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int q = qb * 32 + 8 * j4 + lh4 + e;
#pragma unroll
            for (int n = 0; n < NT; n++)
              if (q < Lq) g.out[oofs + (long)q * g.ldo + n * 32 + lr] = oacc[n][4 * j4 + e] * iv[e];
          }
        }
      } else {
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int q = qb * 32 + acc_row32(r, lh);
            if (q < Lq) g.out[oofs + (long)q * g.ldo + n * 32 + lr] = oacc[n][r];
          }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; s++) { qh[s] = qnh[s]; ql[s] = qnl[s]; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the trailing re-fetch must not outlive the workgroup's LDS allocation
}

int pl_n_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n = prop.multiProcessorCount;
  }
  return n;
}

template <int KT, int NW, bool MAP, int DM>
int launch_pf(const hftt_attn_desc& d, hipStream_t st) {
  using Cfg = PfCfg<KT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(x3p_attn_fwd_kernel<KT, NW, MAP, DM>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) { hftt_set_error("x3p_attn_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  const int cus = pl_n_cus();
  if (cus <= 0) { hftt_set_error("x3p_attn_fwd: device query failed"); return 2; }
  const int per_cu = Cfg::LDS_BYTES <= 53 * 1024 ? 3 : Cfg::LDS_BYTES <= 80 * 1024 ? 2 : 1;      // (<= 96 keys: 49 KB of images, three workgroups of four waves per CU; the launch bound keeps them at <= 170 registers)
  const long items = (long)d.n_seq * d.n_heads;
  const long grid = items < (long)per_cu * cus ? items : (long)per_cu * cus;
  hipLaunchKernelGGL((x3p_attn_fwd_kernel<KT, NW, MAP, DM>), dim3((unsigned)grid), dim3(NW * 64), Cfg::LDS_BYTES, st, d, (int)items);
  HFTT_CHECK_LAUNCH("x3p_attn_fwd");
  return 0;
}
template <int KT, int NW, bool MAP>
int launch_pf3(const hftt_attn_desc& d, hipStream_t st) {
  if (!(d.drop_p > 0.f)) return launch_pf<KT, NW, MAP, 0>(d, st);
  const bool quad_ok = (d.Lk & 3) == 0 && (((uint64_t)d.n_seq * (uint64_t)d.n_heads * (uint64_t)d.Lq * (uint64_t)d.Lk) >> 34) == 0;
  return quad_ok ? launch_pf<KT, NW, MAP, 1>(d, st) : launch_pf<KT, NW, MAP, 2>(d, st);
}
template <int KT, int NW>
int launch_pf2(const hftt_attn_desc& d, hipStream_t st) {
  return d.probs != nullptr ? launch_pf3<KT, NW, true>(d, st) : launch_pf3<KT, NW, false>(d, st);
}

// fp32 [rows, cols] -> f16-pair planes, per 32-column group: one thread per 8 consecutive elements
__global__ __launch_bounds__(256) void x3_to_planes_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, int rows, int cols) {
  const int per_row = cols >> 3;
  const long total = (long)rows * per_row;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int r = (int)(t / per_row), c8 = (int)(t - (long)r * per_row);
    const float* p = src + (long)r * lds + c8 * 8;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    bf16x8 hi, lo;
    x3_split8<X3_F16>(v, hi, lo);
    const int grp = c8 >> 2, w = c8 & 3;                       // 32-column group, 8-element piece inside it
    unsigned char* o = reinterpret_cast<unsigned char*>(dst + (long)r * ldd + grp * 32);
    *reinterpret_cast<bf16x8*>(o + w * 16) = hi;
    *reinterpret_cast<bf16x8*>(o + 64 + w * 16) = lo;
  }
}

#include "x3_attn_bwd.h"

}  // namespace

int hftt_x3p_attn_bwd(const hftt_attn_desc& d, hipStream_t st) {
  const unsigned both = HFTT_ATTN_Q_F16PAIR | HFTT_ATTN_KV_F16PAIR;
  HFTT_REQUIRE((d.io_flags & both) == both && d.dh == 64, "attn_bwd: f16-pair planes need both flags and dh == 64");
  HFTT_REQUIRE(d.ldq % 4 == 0 && d.ldk % 4 == 0 && d.ldv % 4 == 0 && d.q_seq_stride % 4 == 0 && d.k_seq_stride % 4 == 0 && d.v_seq_stride % 4 == 0,
               "attn_bwd: f16-pair planes must be 16-byte aligned");
#ifdef HFTT_X3_ATTN_ABLATE
  hftt_attn_desc da = d;
  da.pad = 0;
  if (const char* e = getenv("HFTT_X3_ATTN_DEBUG")) da.pad = (uint32_t)atoi(e);      // (timing switches of csrc/x3_attn_bwd.h)
  return dispatch_xb<64, true>(da, st);
#else
  return dispatch_xb<64, true>(d, st);
#endif
}

// -1: not this kernel's case (x3_attn.hip handles fp32 operands); otherwise the launch status.  Called by hftt_x3_attn_fwd.
int hftt_x3p_attn_fwd_try(const hftt_attn_desc& d0, hipStream_t st) {
  const unsigned both = HFTT_ATTN_Q_F16PAIR | HFTT_ATTN_KV_F16PAIR;
  if ((d0.io_flags & both) == 0) return -1;
  hftt_attn_desc d = d0;
  d.pad = 0;
#ifdef HFTT_X3_ATTN_ABLATE
  if (const char* e = getenv("HFTT_X3P_DEBUG")) d.pad = (uint32_t)atoi(e);
#endif
  HFTT_REQUIRE((d.io_flags & both) == both, "attn_fwd: q and k / v must both be f16-pair planes (HFTT_ATTN_Q_F16PAIR | HFTT_ATTN_KV_F16PAIR)");
  HFTT_REQUIRE(d.dh == 64, "attn_fwd: f16-pair planes need dh == 64 (got %d)", d.dh);
  HFTT_REQUIRE(d.ldq % 4 == 0 && d.ldk % 4 == 0 && d.ldv % 4 == 0 && d.q_seq_stride % 4 == 0 && d.k_seq_stride % 4 == 0 && d.v_seq_stride % 4 == 0 &&
               ((uintptr_t)d.q & 15) == 0 && ((uintptr_t)d.k & 15) == 0 && ((uintptr_t)d.v & 15) == 0, "attn_fwd: f16-pair planes must be 16-byte aligned");
  const int kt = (d.Lk + 31) / 32, nqb = (d.Lq + 31) / 32;
  if (kt <= 4) {
    HFTT_REQUIRE(nqb <= 4, "attn_fwd (planes): Lq %d needs more than 4 query blocks at Lk %d", d.Lq, d.Lk);
    if (kt <= 1) return launch_pf2<1, 4>(d, st);
    if (kt <= 2) return launch_pf2<2, 4>(d, st);
    if (kt <= 3) return launch_pf2<3, 4>(d, st);
    return launch_pf2<4, 4>(d, st);
  }
  return nqb <= 4 ? launch_pf2<8, 4>(d, st) : launch_pf2<8, 8>(d, st);
}

extern "C" int hftt_x3_to_planes(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t rows, int32_t cols, void* stream) {
  HFTT_REQUIRE(src != nullptr && dst != nullptr && rows > 0 && cols > 0 && cols % 32 == 0, "x3_to_planes: cols must be a multiple of 32 (got %d x %d)", rows, cols);
  HFTT_REQUIRE(lds % 4 == 0 && ldd % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "x3_to_planes: rows must be 16-byte aligned");
  HFTT_REQUIRE(src != dst, "x3_to_planes: not an in-place operation (a thread's 16-byte outputs overlap its neighbours' 32-byte inputs)");
  const long total = (long)rows * (cols / 8);
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(x3_to_planes_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, (long)lds, dst, (long)ldd, rows, cols);
  HFTT_CHECK_LAUNCH("x3_to_planes");
  return 0;
}
