// Attention backward of the split-operand ("x3") mode (gfx950), shared by x3_attn.hip (fp32 operands) and x3_attn_pl.hip (q, k, v as f16-pair
// planes): the kernel template and its dispatch.  Included INSIDE an anonymous namespace, after hftt_common.h / x3_common.h / hftt_host.h.
#pragma once
#ifndef XABL
#ifdef HFTT_X3_ATTN_ABLATE
#define XABL(g, bit) (((g).pad & (bit)) != 0)
#else
#define XABL(g, bit) false
#endif
#endif
constexpr float XB_LOG2E = 1.4426950408889634f;
#ifndef HFTT_XB_SG_A
#define HFTT_XB_SG_A 10         // vector instructions between a dQ step's fragment reads and its first MFMA / between its MFMAs (tools/sweep_xb_sg.sh)
#define HFTT_XB_SG_B 6
#endif
#ifndef HFTT_XB_IL
#define HFTT_XB_IL 1           // 0: the 256-key form without the interleaved dQ product (A/B builds: tools/stamp_x3_attn.sh)
#endif
// -DHFTT_X3_ATTN_STAMPS (tools/stamp_x3_attn.sh): lane 0 of every wave of workgroups 1024 .. 1087 stamps the shader clock around the phases of
// query block 3 into g.probs (unused by the backward) as [item][wave][32] int64 (16, 17: the 100 MHz counter at the item's start and end)
#ifdef HFTT_X3_ATTN_STAMPS
#define XSTAMP(k) do { if (stamp_on) stamp_p[k] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define XSTAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
// f16-pair planes (HFTT_ATTN_*_F16PAIR, dh == 64: x3_attn_pl.hip): every aligned 32-column group of a row = 32 hi halves (64 B), then 32 lo
// halves.  Four consecutive elements 4*c4 .. of one (row, head) = 8 bytes of each; hi + lo is exact in fp32 (at most 23 significant bits)
__device__ __forceinline__ void pl_load4(const float* head_row, int c4, uint2& hi, uint2& lo) {
  const unsigned char* p = reinterpret_cast<const unsigned char*>(head_row) + (c4 >> 3) * 128 + (c4 & 7) * 8;
  hi = *reinterpret_cast<const uint2*>(p);
  lo = *reinterpret_cast<const uint2*>(p + 64);
}
__device__ __forceinline__ float4 pl_vals4(const uint2& hi, const uint2& lo) {
  float h0, h1, h2, h3, l0, l1, l2, l3;
  X3<X3_F16>::unpk(hi.x, h0, h1); X3<X3_F16>::unpk(hi.y, h2, h3);
  X3<X3_F16>::unpk(lo.x, l0, l1); X3<X3_F16>::unpk(lo.y, l2, l3);
  return make_float4(h0 + l0, h1 + l1, h2 + l2, h3 + l3);
}
// eight consecutive elements (one MFMA fragment): 16 bytes of each plane
__device__ __forceinline__ void pl_load8(const float* head_row, int e0, bf16x8& hi, bf16x8& lo) {
  const unsigned char* p = reinterpret_cast<const unsigned char*>(head_row) + (e0 >> 5) * 128 + (e0 & 31) * 2;
  hi = *reinterpret_cast<const bf16x8*>(p);
  lo = *reinterpret_cast<const bf16x8*>(p + 64);
}
__device__ __forceinline__ void pl_vals8(const bf16x8& hi, const bf16x8& lo, float* v) {
  const uint4 h = __builtin_bit_cast(uint4, hi), l = __builtin_bit_cast(uint4, lo);
  const unsigned hh[4] = {h.x, h.y, h.z, h.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    float a0, a1, b0, b1;
    X3<X3_F16>::unpk(hh[q], a0, a1); X3<X3_F16>::unpk(ll[q], b0, b1);
    v[2 * q] = a0 + b0; v[2 * q + 1] = a1 + b1;
  }
}

template <int KT, int DH>
struct XbCfg {
  static constexpr int LKP = KT * 32;
  // LDS images: UNPADDED rows (dh halves) whose 16-byte chunks are XOR-swizzled by row bits, chosen (exhaustive search over linear maps, bank
  // rule of MI355X_MICROARCH.md, LDS) so that every read of the loop is conflict-free: the row reads (ds_read_b128: 16 rows x one chunk per
  // lane group) AND the transposed reads (ds_read_b64_tr_b16: four / eight rows x 64 / 32 bytes per 32 lanes) of the same image.  The
  // padded rows before (Q / dO + 16 bytes, K 192 / 64 bytes, dS + 16 bytes) were 2-way on the transposed reads of Q^T / dO^T and K and on
  // the dS row reads: SQ_LDS_BANK_CONFLICT = 37 % of SQ_LDS_IDX_ACTIVE at dh = 64 (profiles/r04b_attn_bwd_lds.txt).
  static constexpr int RSK = DH;                      // K image (bf16 pair) for dQ: transposed reads only
  static constexpr int RSQ = DH;                      // Q / dO blocks: row reads (scores, dP) and transposed reads (dK, dV)
  static constexpr int RSS = LKP + 16;                // dS block [query][key]: row reads, 32 bytes mod 256 per row
  // chunk ^= swizzle(row).  dh = 64 -- Q / dO: bit 0 <- row bit 1, bit 1 <- row bit 2, bit 2 <- row bits 1 ^ 3;  K: bit 1 <- row bit 1,
  // bit 2 <- row bit 3.   dh = 32 (four chunks per row) -- Q / dO: bits 0, 1 <- row bits 2, 3;  K: bit 1 <- row bit 3
  static __device__ __forceinline__ int qsw(int row) { return DH == 64 ? (((row >> 1) & 3) | ((((row >> 1) ^ (row >> 3)) & 1) << 2)) : ((row >> 2) & 3); }
  static __device__ __forceinline__ int ksw(int row) { return DH == 64 ? ((row & 2) | (((row >> 3) & 1) << 2)) : (((row >> 3) & 1) << 1); }
  static constexpr int R8X = DH == 64 ? 32 : 16;      // what row + 8 (row bit 3) does to a Q / dO column offset
  // element offset (in halves) of column `col` (a multiple of 4) of a row
  static __device__ __forceinline__ int qoff(int row, int col) { return row * RSQ + ((((col >> 3) ^ qsw(row)) << 3) | (col & 7)); }
  static __device__ __forceinline__ int koff(int row, int col) { return row * RSK + ((((col >> 3) ^ ksw(row)) << 3) | (col & 7)); }
  static constexpr int K_PL = LKP * RSK;
  static constexpr int Q_PL = 32 * RSQ;
  static constexpr int S_PL = 32 * RSS;
  // IL (256 keys: one workgroup per CU whatever the size): the dQ product of query block qb - 1 is issued inside the softmax-backward vector
  // phase of block qb (matrix pipe and LDS under vector work; it was a phase of its own behind barrier (h): 150 of 1007 us standalone), which
  // needs the dS block double-buffered.  The smaller forms keep one buffer: 76 KB = two workgroups per CU at 128 keys.
  static constexpr bool IL = KT == 8 && DH == 64 && HFTT_XB_IL != 0;   // (dh = 32: four dQ tiles for eight waves -- the interleaved form measured no gain there)
  static constexpr int NSB = IL ? 2 : 1;
  // Q fp16 (hi, lo) | Q bf16 (hi, lo) | dO bf16 (hi, lo) | row statistics | dS bf16 (hi, lo) x NSB | K image (bf16 hi, lo)
  static constexpr int ELEMS = 2 * K_PL + 6 * Q_PL + NSB * 2 * S_PL;
  static constexpr int LDS_LOOP = ELEMS * 2 + 96 * 4;
  static constexpr int RSE = DH + 4;                  // epilogue patch rows (floats)
  static constexpr int LDS_EPI = KT * 2 * 32 * RSE * 4;
  static constexpr int LDS_BYTES = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
  static constexpr int NTHR = KT * 64;
};

// PL: q, k, v are f16-pair planes (dh == 64): the fp16 halves the score recomputation needs are the stored bytes (the forward's, bit for
// bit); the bf16 pairs of the gradient products are formed from hi + lo
// DM: the dropout form (0 none, 1 one hash per aligned key quad, 2 per element), chosen by the host (as in x3_attn_pl.hip: three forms
// instantiated side by side inside the query-block loop cost registers and scratch)
// PL kernels are PERSISTENT: a workgroup walks items (sequence, head) blockIdx.x, + gridDim.x, ... and requests the next item's K / V rows and
// first query block from inside the current item's last query block (the registers those live in are free there), so the item prologue --
// 13 % of an item at one workgroup per CU, with nothing to overlap it -- is reduced to its conversions.
struct XbBases { const unsigned char *q, *dout, *o, *l; const float *k, *v; };
// Second launch bound = waves per SIMD the registers must allow: the forms of 96 / 128 keys hold two workgroups per CU by their LDS, and
// without the bound hipcc spreads them over 280 - 300 registers -- accumulation registers as spill space -- which halves their occupancy
// (tests/test_kernel_resources.py reads the compiler's resource remarks for exactly this).
template <int KT, int DH, bool PL, int DM>
__global__ __launch_bounds__(KT * 64, (KT == 4 || (KT == 3 && (PL || DH == 32))) ? 2 : 1) void x3_attn_bwd_kernel(const hftt_attn_desc g) {
  constexpr bool PS = PL;
  static_assert(!PL || DH == 64, "f16-pair planes: dh == 64");
  using Cfg = XbCfg<KT, DH>;
  constexpr int EF = X3_F16, EB = X3_BF16;
  constexpr int RSK = Cfg::RSK, RSQ = Cfg::RSQ, RSS = Cfg::RSS, LKP = Cfg::LKP, NTHR = Cfg::NTHR;
  constexpr int K_PL = Cfg::K_PL, Q_PL = Cfg::Q_PL, S_PL = Cfg::S_PL;
  constexpr int KS = DH / 16, NT = DH / 32, F4R = DH / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the small images first: their byte offsets (planes, k-steps, row groups) then fit the 16-bit immediate of the DS instructions; behind
  // the 64 KB K image every read of them paid a v_add for its address
  unsigned short* Qf = reinterpret_cast<unsigned short*>(smem);       // fp16 hi | lo   (scores)
  unsigned short* Qb = Qf + 2 * Q_PL;                                   // bf16 hi | lo   (dK)
  unsigned short* Ob = Qb + 2 * Q_PL;                                   // dO, bf16 hi | lo
  float* lse_s = reinterpret_cast<float*>(Ob + 2 * Q_PL);              // row statistics (in front of the dS images: immediate offsets)
  unsigned short* Sb0 = reinterpret_cast<unsigned short*>(lse_s + 96);  // dS, bf16 hi | lo  (x 2 buffers when IL)
  unsigned short* Kb = Sb0 + Cfg::NSB * 2 * S_PL;                       // bf16 hi | lo (dQ)
  float* delta_s = lse_s + 32;
  float* inv_s = lse_s + 64;
  constexpr bool IL = Cfg::IL;
  // the next block's rows are requested behind the vector phase (IL: they would be live across it together with the dQ fragments; 96
  // keys: 3 sets of prefetch registers per lane do not fit the 256 registers of two workgroups per CU beside that phase)
  constexpr bool LATE_Q = IL || KT == 3;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef HFTT_PRIO_SKEW
  // waves w and w + 4 share a SIMD and run the same phases between the same barriers: a static priority makes the first finish its matrix
  // phase while the second waits, after which one is in vector work while the other multiplies
  if (KT >= 8 && __builtin_amdgcn_readfirstlane(wave) < KT / 2) __builtin_amdgcn_s_setprio(HFTT_PRIO_SKEW);
#endif
  const int lr0 = lane & 31, lh0 = lane >> 5;
  const int gi0 = lane >> 4, qq0 = (lane & 15) >> 2, pp0 = lane & 3;
  int item = blockIdx.x;
  const int n_items = g.n_seq * g.n_heads;
  const int Lq = g.Lq, Lk = g.Lk;
  const float scale = 1.0f / sqrtf((float)DH);
  const float c2 = scale * XB_LOG2E;
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const float keep_inv = g.drop_p > 0.f ? (float)thr * (1.0f / 256.0f) : 1.0f;      // 1 / inv_keep, exactly

  const int nqb = (Lq + 31) / 32;
  long dqofs = 0, sh = 0;                             // (of the current item)

  // Q / dO / O rows (+ row statistics) of the NEXT query block are fetched into registers while the current block is processed
  constexpr int QCNT = (32 * F4R + NTHR - 1) / NTHR;
  float4 pq[QCNT], pdo[QCNT], po[QCNT];
  float2 pl[QCNT];
  // (wave-uniform base pointers + 32-bit byte offsets per lane: one 24-bit multiply per row stride instead of 64-bit address arithmetic
  // per tensor; a sequence's rows span Lq * ld * 4 bytes, far below 2^32)
  auto bases = [&](int it) {
    const int seq_ = it / g.n_heads, head_ = it % g.n_heads;
    const long oofs_ = (long)seq_ * g.o_seq_stride + head_ * DH;
    XbBases b;
    b.q = reinterpret_cast<const unsigned char*>(g.q + (long)seq_ * g.q_seq_stride + head_ * DH);
    b.dout = reinterpret_cast<const unsigned char*>(g.dout + oofs_);
    b.o = reinterpret_cast<const unsigned char*>(g.out + oofs_);
    b.l = reinterpret_cast<const unsigned char*>(g.lse + ((long)seq_ * g.n_heads + head_) * Lq * 2);
    b.k = g.k + (long)seq_ * g.k_seq_stride + head_ * DH;
    b.v = g.v + (long)seq_ * g.v_seq_stride + head_ * DH;
    return b;
  };
  const unsigned ldq_b = (unsigned)g.ldq * 4u, ldo_b = (unsigned)g.ldo * 4u;
  auto qload = [&](const XbBases& b, int qb) {
    int tid_q = tid;
    asm volatile("" : "+v"(tid_q));                    // (row / column offsets are formed per call, not kept in registers across the loop)
#pragma unroll
    for (int u = 0; u < QCNT; u++) {
      const int i = tid_q + NTHR * u;
      const int ic = i < 32 * F4R ? i : 32 * F4R - 1;
      const int row = ic / F4R, cs = ic % F4R;
      const int q = qb * 32 + row;
      const unsigned qc = (unsigned)(q < Lq ? q : Lq - 1);   // clamped address: loads stay unconditional (rows past Lq are zeroed at consumption)
      const unsigned rq = __umul24(qc, ldq_b), ro = __umul24(qc, ldo_b) + (unsigned)cs * 16u;
      if (PL) {                                    // (hi, lo) pairs of the four elements, carried in the same four registers
        const unsigned char* p_ = b.q + (rq + (unsigned)((cs >> 3) * 128 + (cs & 7) * 8));     // (pl_load4's layout)
        const uint2 ph_ = *reinterpret_cast<const uint2*>(p_), pl_ = *reinterpret_cast<const uint2*>(p_ + 64);
        pq[u] = make_float4(__uint_as_float(ph_.x), __uint_as_float(ph_.y), __uint_as_float(pl_.x), __uint_as_float(pl_.y));
      } else {
        pq[u] = *reinterpret_cast<const float4*>(b.q + (rq + (unsigned)cs * 16u));
      }
      pdo[u] = *reinterpret_cast<const float4*>(b.dout + ro);
      po[u] = *reinterpret_cast<const float4*>(b.o + ro);
      pl[u] = *reinterpret_cast<const float2*>(b.l + qc * 8u);
    }
  };

  // ---- this wave's K rows (fp16 pair: the scores) and V rows (bf16 pair: dP) as B-operand fragments (B[k = dh][col = key]).  The K image of
  // the dQ product (row-major, bf16 pair) is written from the same registers: the eight waves' rows are the image (it was loaded a second
  // time, 64 KB per item through the vector memory path) ----
  const int mykey = wave * 32 + lr0;
  const int mykey_c = mykey < Lk ? mykey : Lk - 1;  // clamped address + select: loads stay unconditional
  bf16x8 kfh[KS], kfl[KS], vfh[KS], vfl[KS];         // PL: V's fp16 (hi, lo) as loaded until frag_convert
  float4 kraw[PL ? 1 : KS][2], vraw[PL ? 1 : KS][2];
  auto frag_load = [&](const XbBases& b) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));                      // (lane indices and addresses formed here from the thread index: nothing of them is kept live -- or spilled -- across the query-block loop)
    const int mkey_ = (t_ >> 6) * 32 + (t_ & 31);
    const int mk = mkey_ < Lk ? mkey_ : Lk - 1, lhh = (t_ >> 5) & 1;
#pragma unroll
    for (int s = 0; s < KS; s++) {
      if (PL) {
        pl_load8(b.k + (long)mk * g.ldk, 16 * s + 8 * lhh, kfh[s], kfl[s]);
        pl_load8(b.v + (long)mk * g.ldv, 16 * s + 8 * lhh, vfh[s], vfl[s]);
      } else {
        const float* kp = b.k + (long)mk * g.ldk + 16 * s + 8 * lhh;
        const float* vp = b.v + (long)mk * g.ldv + 16 * s + 8 * lhh;
        kraw[PL ? 0 : s][0] = *reinterpret_cast<const float4*>(kp); kraw[PL ? 0 : s][1] = *reinterpret_cast<const float4*>(kp + 4);
        vraw[PL ? 0 : s][0] = *reinterpret_cast<const float4*>(vp); vraw[PL ? 0 : s][1] = *reinterpret_cast<const float4*>(vp + 4);
      }
    }
  };
  auto frag_convert = [&]() {
    int t_ = tid;
    asm volatile("" : "+v"(t_));                      // (as in frag_load)
    const int mykey = (t_ >> 6) * 32 + (t_ & 31), lh0 = (t_ >> 5) & 1;
#pragma unroll
    for (int s = 0; s < KS; s++) {
      float kv[8], vv[8];
      if (PL) {
        pl_vals8(kfh[s], kfl[s], kv);
        pl_vals8(vfh[s], vfl[s], vv);
      } else {
        const float4 a0 = kraw[PL ? 0 : s][0], a1 = kraw[PL ? 0 : s][1], b0 = vraw[PL ? 0 : s][0], b1 = vraw[PL ? 0 : s][1];
        kv[0] = a0.x; kv[1] = a0.y; kv[2] = a0.z; kv[3] = a0.w; kv[4] = a1.x; kv[5] = a1.y; kv[6] = a1.z; kv[7] = a1.w;
        vv[0] = b0.x; vv[1] = b0.y; vv[2] = b0.z; vv[3] = b0.w; vv[4] = b1.x; vv[5] = b1.y; vv[6] = b1.z; vv[7] = b1.w;
      }
      if (mykey >= Lk) {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        if (PL) { kfh[s] = z; kfl[s] = z; }
#pragma unroll
        for (int e = 0; e < 8; e++) { kv[e] = 0.f; vv[e] = 0.f; }
      }
      if (!PL) x3_split8<EF>(kv, kfh[s], kfl[s]);
      x3_split8<EB>(vv, vfh[s], vfl[s]);
      bf16x8 ih, il;                                  // the image row piece: elements 16 s + 8 lh .. + 8 of key `mykey`
      x3_split8<EB>(kv, ih, il);
      const int ko = Cfg::koff(mykey, 16 * s + 8 * lh0);
      *reinterpret_cast<bf16x8*>(Kb + ko) = ih;
      *reinterpret_cast<bf16x8*>(Kb + K_PL + ko) = il;
    }
  };
  XbBases cur = bases(item);
  frag_load(cur);
  qload(cur, 0);
  __builtin_amdgcn_sched_barrier(0);                  // (nothing below is moved in front of a load)
  f32x16 dKT[NT], dVT[NT];

  // ---- dQ block = dS . K with 16x16 tiles spread over the waves: tile t = wave + KT * (i / KT), key tile ks = i % KT of step i ----
  constexpr int CT = DH / 16, NTILE = 2 * CT, TPW = (NTILE + KT - 1) / KT, TS = TPW * KT;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  f32x4 dq_acc = {0.f, 0.f, 0.f, 0.f};
  // (gi, qq, pp: the caller's opaque copies of the lane indices, see the loop)
  auto dq_step = [&](int i, const unsigned short* Sbuf, int gi, int qq, int pp) {
    const int t = wave_s + KT * (i / KT), ks = i % KT;
    if (TPW * KT == NTILE || t < NTILE) {                   // (wave-uniform)
      const int qh2 = t / CT, ct = t % CT;
      if (ks == 0) { dq_acc[0] = 0.f; dq_acc[1] = 0.f; dq_acc[2] = 0.f; dq_acc[3] = 0.f; }
      const unsigned short* ps = Sbuf + (qh2 * 16 + (lane & 15)) * RSS + 8 * gi + ks * 32;
      const unsigned short* pk = Kb + Cfg::koff(8 * gi + qq, ct * 16 + 4 * pp) + ks * 32 * RSK;   // (the swizzle does not depend on ks, nor on the + 4 rows below)
      const bf16x8 ah = lds_read_b128(ps), al = lds_read_b128(ps + S_PL);
      const bf16x8 bh = join4(lds_read_tr16(pk), lds_read_tr16(pk + 4 * RSK));
      const bf16x8 bl = join4(lds_read_tr16(pk + K_PL), lds_read_tr16(pk + K_PL + 4 * RSK));
      dq_acc = x3_mma16<EB>(ah, al, bh, bl, dq_acc);
    }
  };
  // the finished tile of step i (its last key tile) -> dQ rows of query block qbp (nothing is stored for qbp < 0)
  auto dq_store = [&](int i, int qbp, int gi) {
    const int t = wave_s + KT * (i / KT);
    if ((TPW * KT == NTILE || t < NTILE) && qbp >= 0 && !XABL(g, 32)) {
      const int qh2 = t / CT, ct = t % CT;
      const int q0 = qbp * 32 + qh2 * 16 + gi * 4;
      unsigned char* dq0 = reinterpret_cast<unsigned char*>(g.dq + dqofs);
      const unsigned lddq_b = (unsigned)g.lddq * 4u;
      const unsigned dqo = __umul24((unsigned)q0, lddq_b) + (unsigned)(ct * 16 + (lane & 15)) * 4u;
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (q0 + r < Lq) *reinterpret_cast<float*>(dq0 + (dqo + (unsigned)r * lddq_b)) = dq_acc[r];
    }
  };

  for (;;) {                                          // items of this workgroup
  const int seq = item / g.n_heads, head = item % g.n_heads;
  dqofs = (long)seq * g.dq_seq_stride + head * DH;
  sh = (long)seq * g.n_heads + head;
  const int nxt = item + (int)gridDim.x;
  const bool has_next = PS && nxt < n_items;
  const XbBases nb = bases(has_next ? nxt : item);
#ifdef HFTT_X3_ATTN_STAMPS
  long long* stamp_p = reinterpret_cast<long long*>(g.probs) + ((long)(item - 1024) * KT + wave) * 32;
  const bool stamp_wg = g.probs != nullptr && item >= 1024 && item < 1088 && lane == 0;
  if (stamp_wg) { stamp_p[10] = (long long)__builtin_amdgcn_s_memtime(); stamp_p[16] = (long long)__builtin_amdgcn_s_memrealtime(); }
#endif
  frag_convert();
#pragma unroll
  for (int n = 0; n < NT; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dKT[n][r] = 0.f; dVT[n][r] = 0.f; }
#ifdef HFTT_X3_ATTN_STAMPS
  if (stamp_wg) stamp_p[11] = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (int qb = 0; qb < nqb; qb++) {
    const bool last_qb = qb + 1 == nqb;
#ifdef HFTT_X3_ATTN_STAMPS
    const bool stamp_on = stamp_wg && qb == 3;
    if (stamp_wg && qb == 4) stamp_p[9] = (long long)__builtin_amdgcn_s_memtime();
#endif
    XSTAMP(0);
    unsigned short* Sb = Sb0 + (IL ? (qb & 1) * (2 * S_PL) : 0);                  // this block's dS image
    const unsigned short* Sprev = Sb0 + (IL ? ((qb & 1) ^ 1) * (2 * S_PL) : 0);   // the previous block's (IL)
    // lane-derived indices as values the optimiser cannot see through: every LDS / global address below is then formed where it is used.
    // Left visible, LICM hoists dozens of loop-invariant addresses out of this loop and keeps them live across it (spills at KT = 8, dh = 64).
    int lr = lr0, lh = lh0, gi = gi0, qq = qq0, pp = pp0;
    asm volatile("" : "+v"(lr), "+v"(lh), "+v"(gi), "+v"(qq), "+v"(pp));
    // wave-uniform floats live in vector registers on this ISA (no scalar float ALU): pinned to scalar registers by v_readfirstlane
    const float keep_inv_s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(keep_inv)));
    const float sk_s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(scale * inv_keep)));
    // ---- (a) registers -> LDS: the Q block (both forms) and the dO block, delta = rowsum(dO*O), row statistics ----
    int tid_a = tid;
    asm volatile("" : "+v"(tid_a));
#pragma unroll
    for (int u = 0; u < QCNT; u++) {
      const int i = tid_a + NTHR * u;
      if (i < 32 * F4R) {                           // wave-uniform (32*F4R and NTHR are multiples of 64)
        const int row = i / F4R, cs = i % F4R;
        if (qb * 32 + row >= Lq) { pq[u] = make_float4(0.f, 0.f, 0.f, 0.f); pdo[u] = pq[u]; po[u] = pq[u]; pl[u] = make_float2(0.f, 0.f); }
        uint2 hi, lo;
        if (!XABL(g, 1)) {
        if (PL) {                                     // the stored fp16 pair IS the score operand; its sum feeds the bf16 pair of dK
          hi = make_uint2(__float_as_uint(pq[u].x), __float_as_uint(pq[u].y));
          lo = make_uint2(__float_as_uint(pq[u].z), __float_as_uint(pq[u].w));
          pq[u] = pl_vals4(hi, lo);
        } else {
          x3_split4<EF>(pq[u], hi, lo);
        }
        const int so = Cfg::qoff(row, cs * 4);
        *reinterpret_cast<uint2*>(Qf + so) = hi;
        *reinterpret_cast<uint2*>(Qf + Q_PL + so) = lo;
        x3_split4<EB>(pq[u], hi, lo);
        *reinterpret_cast<uint2*>(Qb + so) = hi;
        *reinterpret_cast<uint2*>(Qb + Q_PL + so) = lo;
        x3_split4<EB>(pdo[u], hi, lo);
        *reinterpret_cast<uint2*>(Ob + so) = hi;
        *reinterpret_cast<uint2*>(Ob + Q_PL + so) = lo;
        }
        float dot = pdo[u].x * po[u].x + pdo[u].y * po[u].y + pdo[u].z * po[u].z + pdo[u].w * po[u].w;
        dot = group_sum<F4R>(dot);
        if (cs == 0) {
          // per-row constants of the softmax backward, with everything that is constant along a row folded in once (one lane per row
          // instead of one multiply per score): with s = 1/sqrt(dh), k = the dropout scale (1 / keep probability),
          //   p_k = P * s * k = 2^((S - max) * c2) * (1/sum * s * k),   dS = P * (M * dPd - delta) * s = p_k * ((kept ? dPd : 0) - delta / k),
          //   Pd = P * M = (kept ? p_k : 0) / s:  the dV accumulators are multiplied by 1/s once, in the epilogue
          // (ONE address, immediate offsets, scalar constants: as three hoisted addresses + two constants in vector registers these five
          // values were spilled and reloaded from scratch, one s_waitcnt vmcnt(0) each, in every query block)
          float* st_ = lse_s + row;
          st_[0] = pl[u].x;                              // the RAW row maximum the forward subtracted
          st_[32] = dot * keep_inv_s;                    // delta_s
          st_[64] = pl[u].y * sk_s;                      // inv_s
        }
      }
    }
    XSTAMP(1);
    __syncthreads();   // (b)
    XSTAMP(2);
    // (IL: the prefetch registers would be live across the vector phase that now also holds the dQ fragments -- requested behind it instead,
    // one matrix phase and a barrier ahead of their use)
    if (!LATE_Q) {
      if (!last_qb) qload(cur, qb + 1);
      else if (has_next) qload(nb, 0);              // the next item's first query block
    }

    // ---- (c) S tile and (d) dP tile: rows = queries (registers), column = this lane's key ----
    f32x16 sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; r++) { sacc[r] = 0.f; pacc[r] = 0.f; }
    if (!XABL(g, 2)) {
      // the fragments of k-step s + 1 are requested before the six MFMAs of step s (one step ahead: with both waves of a SIMD in this phase
      // together nobody else covers the LDS round trip; all KS steps up front spill at KT = 8, dh = 64)
      // k-step s = chunks 2s + lh of the row: the swizzle is an XOR, so step s is at off0 ^ 16s (row * RSQ has those bits clear when swizzled)
      const int off0 = Cfg::qoff(lr, 8 * lh);
      auto offs = [&](int s) { return off0 ^ (16 * s); };
      bf16x8 fq[2], fo[2];
      fq[0] = lds_read_b128(Qf + off0); fq[1] = lds_read_b128(Qf + Q_PL + off0);
      fo[0] = lds_read_b128(Ob + off0); fo[1] = lds_read_b128(Ob + Q_PL + off0);
#pragma unroll
      for (int s = 0; s < KS; s++) {
        bf16x8 nq[2], no[2];
        if (s + 1 < KS) {
          const int off = offs(s + 1);
          nq[0] = lds_read_b128(Qf + off); nq[1] = lds_read_b128(Qf + Q_PL + off);
          no[0] = lds_read_b128(Ob + off); no[1] = lds_read_b128(Ob + Q_PL + off);
        }
        // the forward's order of partial sums (it multiplied K as the A operand): K_lo.Q_hi, then K_hi.Q_lo, then K_hi.Q_hi
        sacc = X3<EF>::mma(fq[0], kfl[s], sacc);
        sacc = X3<EF>::mma(fq[1], kfh[s], sacc);
        sacc = X3<EF>::mma(fq[0], kfh[s], sacc);
        pacc = x3_mma<EB>(fo[0], fo[1], vfh[s], vfl[s], pacc);
        if (s + 1 < KS) { fq[0] = nq[0]; fq[1] = nq[1]; fo[0] = no[0]; fo[1] = no[1]; }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    XSTAMP(3);
    // the next item's K / V rows: this item's fragments had their last use above
    if (PS && last_qb && has_next) frag_load(nb);
    const bool key_ok = mykey < Lk;
    const bool pad_wave = (wave * 32 + 32) > Lk;         // (wave-uniform) this wave's key tile has padding columns
    // The 16 registers of a lane are 16 query rows: rows 8j + 4*lh + {0,1,2,3} for j = r >> 2 -> one 16-byte LDS read per statistic and j.
    // On exit sacc = (kept ? p_k : 0) (the B operand of dV, scaled by s: see the staging above), pacc = dS * s (for dK, dQ).
    {
      constexpr bool DROP = DM != 0, PAIR = DM == 1;
      const long row0 = sh * Lq + (long)qb * 32 + 4 * lh;                  // element row of register 0
      const uint64_t ebase = (uint64_t)(row0 * (long)Lk + mykey);
      // PAIR (quad form): lanes 4i .. 4i+3 hold the four keys of one hash quad and registers 4j .. 4j+3 four adjacent rows.  Lane 4i+a hashes
      // the quad of row(4j + a); a DPP quad broadcast hands every lane each row's word, of which it takes its own key's byte.
      const int sub = lane & 3;
      const uint32_t fsh = 8u * (uint32_t)sub;
      const uint64_t hk = hftt_hash_key(g.drop_seed, g.drop_site);
      const uint32_t quarter = (uint32_t)(Lk >> 2);
      const uint32_t qlo = (uint32_t)(row0 + sub) * quarter + (uint32_t)(mykey >> 2);
#pragma unroll
      for (int j4 = 0; j4 < 4; j4++) {
        // IL: a quarter of the previous block's dQ steps in front of every quarter of the vector work (block 0 multiplies whatever the LDS
        // holds and stores nothing: no branch, so the steps stay inside this scheduling region)
        if (IL && TPW == 1 && !XABL(g, 16) && !XABL(g, 64)) {
#pragma unroll
          for (int i = j4 * TS / 4; i < (j4 + 1) * TS / 4; i++) dq_step(i, Sprev, gi, qq, pp);
        }
        const float4 m4 = *reinterpret_cast<const float4*>(lse_s + 8 * j4 + 4 * lh);
        const float4 i4 = *reinterpret_cast<const float4*>(inv_s + 8 * j4 + 4 * lh);
        const float4 d4 = *reinterpret_cast<const float4*>(delta_s + 8 * j4 + 4 * lh);
        const float rs_m[4] = {m4.x, m4.y, m4.z, m4.w}, rs_i[4] = {i4.x, i4.y, i4.z, i4.w}, rs_d[4] = {d4.x, d4.y, d4.z, d4.w};
        uint32_t wq[4] = {0u, 0u, 0u, 0u};
        if (DROP && PAIR) {
          const uint32_t w = hftt_hash_mix(hk, qlo + (uint32_t)(8 * j4) * quarter, 0u);      // row(4*j4) - row(0) = 8*j4 rows
          wq[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x00, 0xF, 0xF, true);
          wq[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x55, 0xF, 0xF, true);
          wq[2] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xAA, 0xF, 0xF, true);
          wq[3] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xFF, 0xF, 0xF, true);
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int r = 4 * j4 + e;
          float p = __builtin_amdgcn_exp2f((sacc[r] - rs_m[e]) * c2) * rs_i[e];      // p_k
          float pd = p, dp = pacc[r];
          if (DROP) {
            bool kept;
            if (PAIR) kept = ((wq[e] >> fsh) & 0xFFu) < thr;
            else kept = hftt_keep(g.drop_seed, g.drop_site, ebase + (uint64_t)((e + 8 * j4) * Lk), thr);
            pd = kept ? p : 0.f;                                // one decision, two selects
            dp = kept ? dp : 0.f;
          }
          sacc[r] = pd;
          pacc[r] = p * (dp - rs_d[e]);
        }
        if (IL && TPW == 1) {
          // issue order of this quarter: per dQ step its six fragment reads, vector work while they travel, then its three (dependent,
          // 16-cycle) MFMAs with vector instructions between them; what is left of the quarter's vector work follows
#pragma unroll
          for (int i = 0; i < TS / 4; i++) {
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, HFTT_XB_SG_A, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, HFTT_XB_SG_B, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, HFTT_XB_SG_B, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, HFTT_XB_SG_B, 0);
          }
        }
      }
      XSTAMP(4);
      // padding columns (keys past Lk: zero K rows, so S = 0 and p != 0): both products carry the factor p, so they are cleared here, by the
      // one wave that has such columns, instead of by a select per score in every wave
      if (pad_wave) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          sacc[r] = key_ok ? sacc[r] : 0.f;
          pacc[r] = key_ok ? pacc[r] : 0.f;
        }
      }
    }
    // (both behind the vector phase and its padding branch: a branch in front of it would end the scheduling region the dQ steps are placed in)
    if (LATE_Q) qload(last_qb ? nb : cur, last_qb ? 0 : qb + 1);   // (unconditional; in the last block the next item's first rows, or -- no next item -- this item's again)
    if (IL && TPW == 1 && !XABL(g, 16) && !XABL(g, 64)) dq_store(TS - 1, qb - 1, gi);
    XSTAMP(5);
    // ---- (e) dV^T += dO^T . Pd   (f) dK^T += Q^T . dS   (g) dS -> LDS ----
    // First ALL the vector work of the block (four bf16 splits; the dS copy for dQ from the same packed halves), then ONE matrix phase of 24
    // MFMAs whose transposed fragment reads run one step ahead.  (Interleaved per half tile, each group of eight reads was waited for in
    // full right before its six MFMAs, four times per block, with both waves of a SIMD in the same phase: this block cost 293 us of the
    // kernel's 964 for 768 matrix cycles and ~130 vector instructions per wave.)
    if (!XABL(g, 8)) {
      bf16x8 ph[2], plo[2], shh[2], shl[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        float pv[8], sv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { pv[e] = sacc[8 * s2 + e]; sv[e] = pacc[8 * s2 + e]; }
        x3_split8<EB>(pv, ph[s2], plo[s2]);
        x3_split8<EB>(sv, shh[s2], shl[s2]);
      }
      // (g) dS -> LDS [query][key], bf16 pair: word w of shh / shl = registers 2rp, 2rp + 1 (rp = 4*s2 + w), two query rows of this lane's
      // key.  Lanes 2i / 2i+1 hold adjacent keys: one packed pair (ds_write_b32) per register pair
      if (!XABL(g, 16)) {
        const bool odd = lane & 1;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          const uint4 hw = __builtin_bit_cast(uint4, shh[s2]), lw = __builtin_bit_cast(uint4, shl[s2]);
          const unsigned hws[4] = {hw.x, hw.y, hw.z, hw.w}, lws[4] = {lw.x, lw.y, lw.z, lw.w};
#pragma unroll
          for (int w = 0; w < 4; w++) {
            const int rp = 4 * s2 + w;
            const int o = acc_row32(2 * rp + (odd ? 1 : 0), lh) * RSS + wave * 32 + (lr & ~1);
            *reinterpret_cast<unsigned*>(Sb + o) = packed_rows_to_cols(hws[w], odd);
            *reinterpret_cast<unsigned*>(Sb + S_PL + o) = packed_rows_to_cols(lws[w], odd);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      XSTAMP(6);
      // steps (s2, n): A operands dO^T and Q^T of rows 16*s2 .. by transposed reads, B operands the packed halves above
      // this lane's element (row 4*lh + qq, column 16*(gi & 1) + 4*pp) once; steps and the second row group (+ 8: swizzle bit 2 flips) are
      // constant row offsets and one XOR of the 32-column bit
      const int tb0 = Cfg::qoff(4 * lh + qq, 16 * (gi & 1) + 4 * pp);
      auto frag = [&](int s2, int n, bf16x8 (&f)[4]) {
        const int o0 = 16 * s2 * RSQ + (tb0 ^ (32 * n));
        const int o1 = (16 * s2 + 8) * RSQ + (tb0 ^ (32 * n) ^ Cfg::R8X);
        f[0] = join4(lds_read_tr16(Ob + o0), lds_read_tr16(Ob + o1));
        f[1] = join4(lds_read_tr16(Ob + Q_PL + o0), lds_read_tr16(Ob + Q_PL + o1));
        f[2] = join4(lds_read_tr16(Qb + o0), lds_read_tr16(Qb + o1));
        f[3] = join4(lds_read_tr16(Qb + Q_PL + o0), lds_read_tr16(Qb + Q_PL + o1));
      };
      constexpr int NSTEP = 2 * NT;                       // step = s2 * NT + n
      bf16x8 fc[4], fn[4];
      frag(0, 0, fc);
#pragma unroll
      for (int step = 0; step < NSTEP; step++) {
        const int s2 = step / NT, n = step % NT;
        if (step + 1 < NSTEP) frag((step + 1) / NT, (step + 1) % NT, fn);
        dVT[n] = x3_mma<EB>(fc[0], fc[1], ph[s2], plo[s2], dVT[n]);
        dKT[n] = x3_mma<EB>(fc[2], fc[3], shh[s2], shl[s2], dKT[n]);
        if (step + 1 < NSTEP) { fc[0] = fn[0]; fc[1] = fn[1]; fc[2] = fn[2]; fc[3] = fn[3]; }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    XSTAMP(7);
    __syncthreads();   // (h)
    XSTAMP(8);

    // ---- (i) dQ block = dS . K (the forms that do not interleave it: see (d)) ----
    if (!(IL && TPW == 1) && !XABL(g, 16) && !XABL(g, 64)) {
#pragma unroll
      for (int tw = 0; tw < TPW; tw++) {
#pragma unroll 2
        for (int ks = 0; ks < KT; ks++) dq_step(tw * KT + ks, Sb, gi, qq, pp);
        dq_store(tw * KT + KT - 1, qb, gi);
      }
    }
    // no barrier needed here: the next iteration's staging touches only Qf / Qb / Ob / statistics, which no wave reads in (i);
    // barrier (b) of the next iteration orders (i) before the next (g).  IL: block qb's dS image is read during block qb + 1 (behind (h))
    // and rewritten by block qb + 2 (behind the next (h)).
  }
#ifdef HFTT_X3_ATTN_STAMPS
  if (stamp_wg) stamp_p[12] = (long long)__builtin_amdgcn_s_memtime();
#endif
  if (IL && TPW == 1 && !XABL(g, 16) && !XABL(g, 64)) {       // the last block's dQ (barrier (h) of the last iteration is behind us)
    const unsigned short* Slast = Sb0 + ((nqb - 1) & 1) * (2 * S_PL);
    int gi = gi0, qq = qq0, pp = pp0;
    asm volatile("" : "+v"(gi), "+v"(qq), "+v"(pp));           // (per item: see the loop)
#pragma unroll 2
    for (int ks = 0; ks < KT; ks++) dq_step(ks, Slast, gi, qq, pp);
    dq_store(TS - 1, nqb - 1, gi);
  }

#ifdef HFTT_X3_ATTN_STAMPS
  if (stamp_wg) stamp_p[13] = (long long)__builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: dK, dV (this wave's 32 keys) leave through LDS as whole row segments (DH fp32 = 128 / 256 bytes per key row) ----
  constexpr int RSE = Cfg::RSE;
  const float rscale = sqrtf((float)DH);                       // dV was accumulated from probabilities scaled by 1/sqrt(dh) (8: exact at dh = 64)
  __syncthreads();                                             // every wave is done with the loop's LDS images
  int lane_e = lane, lr_e = lr0, lh_e = lh0;
  asm volatile("" : "+v"(lane_e), "+v"(lr_e), "+v"(lh_e));     // (the store addresses are formed per item, not kept -- spilled -- across the item loop)
  float* ek = reinterpret_cast<float*>(smem) + wave_s * (2 * 32 * RSE);
  float* ev = ek + 32 * RSE;
#pragma unroll
  for (int n = 0; n < NT; n++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int dh0 = n * 32 + 8 * c + 4 * lh_e;
      *reinterpret_cast<float4*>(ek + lr_e * RSE + dh0) = make_float4(dKT[n][4 * c], dKT[n][4 * c + 1], dKT[n][4 * c + 2], dKT[n][4 * c + 3]);
      *reinterpret_cast<float4*>(ev + lr_e * RSE + dh0) = make_float4(dVT[n][4 * c] * rscale, dVT[n][4 * c + 1] * rscale, dVT[n][4 * c + 2] * rscale, dVT[n][4 * c + 3] * rscale);
    }
  __syncthreads();
  constexpr int CPR = DH / 4;                                  // 16-byte chunks per row
  constexpr int RPP = 64 / CPR;                                // rows per pass of the wave
  float* dkp = g.dk + (long)seq * g.dk_seq_stride + head * DH;
  float* dvp = g.dv + (long)seq * g.dv_seq_stride + head * DH;
#pragma unroll
  for (int ps = 0; ps < 32 / RPP; ps++) {
    const int row = ps * RPP + lane_e / CPR, ch = lane_e % CPR;
    const int key = wave_s * 32 + row;
    if (key < Lk) {
      *reinterpret_cast<float4*>(dkp + (long)key * g.lddk + ch * 4) = *reinterpret_cast<const float4*>(ek + row * RSE + ch * 4);
      *reinterpret_cast<float4*>(dvp + (long)key * g.lddv + ch * 4) = *reinterpret_cast<const float4*>(ev + row * RSE + ch * 4);
    }
  }
#ifdef HFTT_X3_ATTN_STAMPS
  if (stamp_wg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp_p[14] = (long long)__builtin_amdgcn_s_memtime(); stamp_p[17] = (long long)__builtin_amdgcn_s_memrealtime(); }
#endif
  if (!has_next) break;
  item = nxt;
  cur = nb;
  __syncthreads();                                    // the patches above alias the images the next item writes
  }
}

template <int KT, int DH, bool PL, int DM>
int launch_xb(const hftt_attn_desc& d, hipStream_t st) {
  using Cfg = XbCfg<KT, DH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(x3_attn_bwd_kernel<KT, DH, PL, DM>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) { hftt_set_error("x3_attn_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 2; }
    attr_set = true;
  }
  long grid = (long)d.n_seq * d.n_heads;
  if (PL) {                                           // persistent: as many workgroups as the device holds at once
    static int resident = 0;
    if (resident == 0) {
      int dev = 0, per_cu = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(x3_attn_bwd_kernel<KT, DH, PL, DM>), Cfg::NTHR, Cfg::LDS_BYTES) != hipSuccess || per_cu < 1) {
        hftt_set_error("x3_attn_bwd: device / occupancy query failed");
        return 2;
      }
      resident = per_cu * prop.multiProcessorCount;
    }
    if (grid > resident) grid = resident;
  }
  hipLaunchKernelGGL((x3_attn_bwd_kernel<KT, DH, PL, DM>), dim3((unsigned)grid), dim3(Cfg::NTHR), Cfg::LDS_BYTES, st, d);
  HFTT_CHECK_LAUNCH("x3_attn_bwd");
  return 0;
}
template <int KT, int DH, bool PL>
int launch_xb3(const hftt_attn_desc& d, hipStream_t st) {
  if (!(d.drop_p > 0.f)) return launch_xb<KT, DH, PL, 0>(d, st);
  const bool quad_ok = (d.Lk & 3) == 0 && (((uint64_t)d.n_seq * (uint64_t)d.n_heads * (uint64_t)d.Lq * (uint64_t)d.Lk) >> 34) == 0;
  return quad_ok ? launch_xb<KT, DH, PL, 1>(d, st) : launch_xb<KT, DH, PL, 2>(d, st);
}
template <int DH, bool PL>
int dispatch_xb(const hftt_attn_desc& d, hipStream_t st) {
  const int kt = (d.Lk + 31) / 32;
  // the kernel addresses a sequence's rows with 32-bit byte offsets formed by a 24-bit multiply
  if (d.ldq <= 0 || d.ldo <= 0 || d.ldq >= (1 << 22) || d.ldo >= (1 << 22) || d.lddq <= 0 || d.lddq >= (1 << 22) || (int64_t)d.Lq * d.lddq >= (1ll << 29) || (int64_t)d.Lq * d.ldq >= (1ll << 29) || (int64_t)d.Lq * d.ldo >= (1ll << 29)) {
    hftt_set_error("x3_attn_bwd: row strides too large for 32-bit row offsets (Lq %d, ldq %lld, ldo %lld)", d.Lq, (long long)d.ldq, (long long)d.ldo);
    return 1;
  }
  if (kt <= 1) return launch_xb3<1, DH, PL>(d, st);
  if (kt <= 2) return launch_xb3<2, DH, PL>(d, st);
  if (kt <= 3) return launch_xb3<3, DH, PL>(d, st);
  if (kt <= 4) return launch_xb3<4, DH, PL>(d, st);
  return launch_xb3<8, DH, PL>(d, st);
}

