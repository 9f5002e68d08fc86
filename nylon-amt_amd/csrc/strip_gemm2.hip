// Strip kernels, persistent software-pipelined form ("v2").  Same contract, operand layouts and arithmetic as csrc/strip_gemm.hip
// (which stays as the general form: fp32 storage, gates, K = 128, ragged M); what changes is WHEN memory moves.
//
// Why.  In the one-block-per-workgroup form every workgroup runs load -> MFMA -> store, all workgroups of the chip do so in lockstep
// (in-kernel clock stamps, tools/stamp_strip.py: x-load burst 16k cycles at the HBM limit, MFMA 16k cycles with HBM idle, store burst
// 10k cycles at the HBM limit, per pass), and contention re-synchronises any stagger: HBM is busy 46 % of the time, the matrix pipe
// 24 %.  Here one workgroup per CU (4 waves, one per SIMD, up to 512 registers) walks many 128-token blocks and keeps memory traffic
// flat: while block b is multiplied, the activations of block b+1 are prefetched into a second register set and the results of the
// previous pass are stored -- a fixed few loads and stores per ring slot, in between the MFMAs.
//
// Bookkeeping.  The ring fills (LDS-DMA) and the stores are issued from inline asm, the prefetch loads are ordinary 16-byte loads (one
// global_load_dwordx4 each).  `issued` counts all of them, a mark is taken for each ring fill, and s_waitcnt vmcnt(issued - mark)
// retires that fill (vmcnt retires in issue order).  Only the wait + barrier at the top of a slot carries a memory clobber: inside
// a slot the fill pieces and stores carry none, so the LDS fragment reads are free to be scheduled across them -- and so are the
// slot's prefetch loads, which is why a fill's mark is taken at the END of its slot's region (everything in the region counts as
// older than the fill: stricter, never weaker).  An instruction the kernel does not count (the two statistics stores, a scratch
// access) can likewise only make a wait stricter.  The prefetch loads are NOT asm: a register with an asm load in flight is just a register to the
// allocator, and under pressure it copied / reused such registers before the data had landed (garbage operands, and a memory fault
// where the reused register held an address).  hipcc's own wait for them is coarser than needed (it cannot see the asm operations
// in flight and waits for those too), which costs about one DMA latency per block.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "hftt_common.h"
#include "hftt_host.h"
#include "strip_internal.h"
#include "../../include/hftt_hip.h"

// Timing switches / clock stamps (HFTT_STRIP2_DEBUG, results garbage) exist in the ablation and stamp builds only (-DHFTT_STRIP2_ABLATE,
// -DHFTT_STRIP_STAMPS): as run-time tests inside the slot loops they cost ~1 % of the step.
#if defined(HFTT_STRIP2_ABLATE) || defined(HFTT_STRIP_STAMPS)
#define S2DBG(g, bit) (((g).pad & (bit)) != 0)
#define S2PAD(g) ((g).pad)
#else
#define S2DBG(g, bit) false
#define S2PAD(g) 0
#endif
// ablation builds: HFTT_STRIP2_DEBUG bit 512 raises the wave priority for the MFMA runs of the two-workgroups-per-CU form (measured: DESIGN section 5)
#if defined(HFTT_STRIP2_ABLATE)
#define S2PRIO(g) (((g).pad & 512) != 0)
#else
#define S2PRIO(g) false
#endif

// Compile-time timing switches of the fused block (tools/ablate_mlp2_ct.sh builds one library per mask; results garbage): unlike the run-time
// switches above they leave no branch in the slot bodies, so the schedule of what remains is the product's.
//   1 no ring fills   2 no slot barrier   4 no next-block row prefetch   8 no fragment reads (one fragment feeds a slot's 16 MFMAs)
//   16 no slot wait (s_waitcnt vmcnt)
#ifndef HFTT_MLP2_CT
#define HFTT_MLP2_CT 0
#endif

namespace {

#include "strip_pipe.h"

// The first form of the pipeline state (exact waits from an issue counter and a mark per fill, no fetch past the end of the stream): still
// used by the fused two-GEMM kernel, whose register allocation is at the limit (the always-fill form below tipped it into 132 bytes of scratch).
struct PipeDyn {
  const unsigned short* w;      // strip-packed weight stream of ONE block (S slots)
  int S;                        // slots per block
  long fill_left;               // slots this workgroup still has to fetch (its blocks x S at the start)
  int fill_pos;                 // stream position (0 .. S-1) of the next slot to fetch
  unsigned ring;
  int wave, lane;
  int issued;                   // asm-issued vector-memory instructions so far
  int fill_mark[NSLOT];

  // fetch the next slot of the stream into ring buffer BUF (static: slot number % NSLOT); wave w moves fragments 4w .. 4w+3
  template <int BUF>
  __device__ __forceinline__ void fill() {
    if (fill_left <= 0) return;
    const unsigned short* src = w + ((long)fill_pos * 16 + wave * 4) * 512 + lane * 8;
    const unsigned dst = ring + (unsigned)BUF * SLOT_BYTES + (unsigned)wave * 4096u;
    glds16x4(src, dst);
    issued += 4;
    fill_mark[BUF] = issued;
    fill_left--;
    fill_pos = (fill_pos + 1 == S) ? 0 : fill_pos + 1;
  }
  // top of slot BUF: its DMA has landed in every wave and the buffer consumed one slot ago is free; the refill of that buffer is then
  // issued in four pieces between the slot's MFMA groups (fill_piece), and fill_close takes the mark
  bool filling, closing;
  bool nobar;                   // timing experiments only (HFTT_STRIP2_DEBUG bit 1; bit 0 drops the ring fills): results are garbage
  const unsigned short* fsrc;
  template <int BUF>
  __device__ __forceinline__ void begin_slot() {
    // The refill issued during the previous slot gets its mark HERE, at the end of that slot's region: every operation of the region
    // (prefetch loads float inside it, between the memory-clobbering waits) then counts as older than the refill -- the wait can only
    // be stricter than necessary, never weaker.
    if (closing) {
      fill_mark[(BUF + FILL_AHEAD - 1) & (NSLOT - 1)] = issued;      // previous slot = BUF - 1, its refill went to (BUF - 1 + 3) % 4
      closing = false;
    }
    // steady state: the two regions since the mark issued at least a 4-piece refill each, so vmcnt(8) is never weaker than
    // vmcnt(n); the branch tree is for the tail of the stream only (no more refills, n shrinks)
    const int n = issued - fill_mark[BUF];
    if (n >= 8) HFTT_WAITVM(8); else wait_vmcnt_dyn(n > 0 ? n : 0);
    if (!nobar) __builtin_amdgcn_s_barrier();
    filling = fill_left > 0;
    fsrc = w + ((long)fill_pos * 16 + wave * 4) * 512 + lane * 8;
  }
  template <int BUF, int I>
  __device__ __forceinline__ void fill_piece() {      // BUF: the slot being consumed; the refill goes to (BUF + 3) % 4 (all four pieces at I == 0)
    if (I == 0 && filling) {
      glds16x4(fsrc, ring + (unsigned)((BUF + FILL_AHEAD) & (NSLOT - 1)) * SLOT_BYTES + (unsigned)wave * 4096u);
      issued += 4;
    }
  }
  template <int BUF>
  __device__ __forceinline__ void fill_close() {
    if (filling) {
      closing = true;
      fill_left--;
      fill_pos = (fill_pos + 1 == S) ? 0 : fill_pos + 1;
    }
  }
};

// per-wave pipeline state: the weight ring.  The stream is fetched three slots ahead and ALWAYS: past the workgroup's last slot the fetch
// simply wraps to the head of the stream (three harmless extra slots, drained before the kernel ends).  With a four-piece refill in
// every slot, "at least eight vector-memory instructions were issued after the fill of the slot about to be consumed" holds at every
// slot boundary, so the wait is the constant s_waitcnt vmcnt(8) (vmcnt retires in issue order; the prefetch loads and stores issued in
// between only make it stricter).  The first form kept an issue counter, a mark per fill and a branch tree for exact waits at the stream's
// tail: ~25 scalar instructions per slot, each ~8 cycles with one wave per SIMD (tools/stamp_linear2.py) -- more than the precision bought.
struct Pipe {
  const unsigned short* w;      // strip-packed weight stream of ONE block (S slots)
  int S;                        // slots per block
  int fill_pos;                 // stream position (0 .. S-1) of the next slot to fetch
  unsigned ring;
  int wave, lane;
  int issued;                   // (unused; kept so that the call sites read the same)
  bool nobar, nofill;           // timing experiments only (HFTT_STRIP2_DEBUG bits 1 / 0): results are garbage

  __device__ __forceinline__ const unsigned short* src_of(int pos) const { return w + ((long)pos * 16 + wave * 4) * 512 + lane * 8; }
  __device__ __forceinline__ void advance() { fill_pos = (fill_pos + 1 == S) ? 0 : fill_pos + 1; }
  // prologue: fetch the next slot of the stream into ring buffer BUF; wave w moves fragments 4w .. 4w+3
  template <int BUF>
  __device__ __forceinline__ void fill() {
    if (!nofill) glds16x4(src_of(fill_pos), ring + (unsigned)BUF * SLOT_BYTES + (unsigned)wave * 4096u);
    advance();
  }
  // top of slot BUF: its DMA has landed in every wave and the buffer consumed one slot ago is free
  template <int BUF>
  __device__ __forceinline__ void begin_slot() {
    if (!(HFTT_MLP2_CT & 16)) HFTT_WAITVM(8);
    if (!(HFTT_MLP2_CT & 2) && !nobar) __builtin_amdgcn_s_barrier();
  }
  template <int BUF, int I>
  __device__ __forceinline__ void fill_piece() {      // BUF: the slot being consumed; the refill goes to (BUF + 3) % 4 (all four pieces at I == 0)
    if (I == 0 && !(HFTT_MLP2_CT & 1) && !nofill) glds16x4(src_of(fill_pos), ring + (unsigned)((BUF + FILL_AHEAD) & (NSLOT - 1)) * SLOT_BYTES + (unsigned)wave * 4096u);
  }
  template <int BUF>
  __device__ __forceinline__ void fill_close() { advance(); }
  // before the kernel ends: nothing may still be on its way into LDS (the extra slots fetched past the end)
  __device__ __forceinline__ void drain() { HFTT_WAITVM(0); }
};

// ---------------------------------------------------------------------------------------------------------------------
// C = epi(x . Wl^T + bias), all tensors bf16, K % 256 == 0, N % 256 == 0; LN: N == 256 with dropout / residual / LayerNorm
// ---------------------------------------------------------------------------------------------------------------------
// PASSES = N / 256, KCH = K / 256 are template parameters: with run-time trip counts the allocator shuffled whole register sets
// through scratch at the step boundaries (152 spilled registers in the plain form).
// STP (round 5): the results of a pass go through a wave-private LDS patch on their way into the pending registers, which then hold whole-line
// pieces (piece i: tile pair i >> 2, rows (lane >> 3) + 8 (i & 3), 16-byte chunk lane & 7) instead of the lane's own row pieces -- same registers,
// same deferred schedule (two pieces per slot), but every store instruction writes 8 complete 128-byte lines (strip_pipe.h: patch_put).
template <bool LN, int PASSES, int KCH, bool HR, bool STP>
__global__ __launch_bounds__(256, 1) void strip_linear2_kernel(const hftt_strip_desc g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  constexpr int passes = PASSES;
  constexpr int steps = passes * KCH;               // 8-slot steps per block
  const long nblk = ((long)g.M + 127) / 128;
  const long my_blocks = (nblk - (long)blockIdx.x + gridDim.x - 1) / gridDim.x;
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // bias[N] | gamma[256] | beta[256]
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(g.x);
  unsigned short* cb = reinterpret_cast<unsigned short*>(g.C);
  unsigned short* preb = reinterpret_cast<unsigned short*>(g.pre_ln_out);
  const unsigned short* rb = reinterpret_cast<const unsigned short*>(g.residual);
  const bool relu = g.flags & HFTT_SL_RELU;
  const bool plain = !relu && g.out_scale == 1.0f;
  constexpr bool has_res = HR;                      // residual rows are prefetched into the pending registers (a template parameter: no per-slot test)
  unsigned char* patchC = smem + RING_BYTES + 4 * (g.N + 512) + (STP ? wave * 2 * PATCH_BYTES : 0);      // results | pre-LayerNorm rows
  unsigned char* patchP = patchC + PATCH_BYTES;
  const unsigned short* pend_base = cb;             // STP: wave-uniform base of the deferred results (first row of the wave, first column of the pass)

  Pipe P;
  P.w = g.w; P.S = steps * 8; P.nofill = S2DBG(g, 1); P.fill_pos = 0; P.nobar = S2DBG(g, 2);
  P.ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  P.wave = wave; P.lane = lane; P.issued = 0;

  auto tok_of = [&](long blk) { const long t = blk * 128 + wave * 32 + j; return t < g.M ? t : (long)g.M - 1; };
#ifdef HFTT_STRIP_STAMPS
  if (!LN && S2DBG(g, 4) && tid == 0) (reinterpret_cast<unsigned long long*>(g.ln_mean) + (long)blockIdx.x * 40)[34] = __builtin_amdgcn_s_memtime();
#endif

  // ---- prologue: parameters to LDS (compiler loads), first block's activations, first ring slots
  for (int i = tid; i < g.N; i += 256) prm[i] = g.bias != nullptr ? g.bias[i] : 0.f;
  if (LN) { prm[g.N + tid] = g.ln_gamma[tid]; prm[g.N + 256 + tid] = g.ln_beta[tid]; }
  // xf: this step's activations.  xn: the next step's, in flight.  pend: results of the last finished pass waiting to be stored, two
  // pieces per slot; once a piece is on its way its registers receive the residual rows the NEXT epilogue needs (one set for both).
  u4v xf[16], xn[16], pend[16];
  // STP: the strip rows arrive as whole-line pieces (xn[4 p + k]: tile pair p, rows (lane >> 3) + 8 k, chunk lane & 7) from a wave-uniform base
  // (M % 32 == 0: a wave's 32 rows are all valid or all past the end -- then the last 32 rows are read again, harmlessly) and are turned into
  // the lane's own row pieces through the result patch when a step starts (patch_put_lines / patch_get_rows)
  auto wave_rows = [&](long blk) { const long r0 = blk * 128 + wave * 32; const long rc = r0 < (long)g.M ? r0 : (long)g.M - 32; return xb + rc * g.ldx; };
  // whole-line loads where the launch is load-heavy (K >= 512: dX of the fused projections, same box 187-197 -> 176 us); at K = 256 the strip
  // is loaded once per block and the LDS round trip costs more than it saves (QKV 174 -> 180 us).  HFTT_LINEAR2_LINES=0 / 1 forces either (A/B).
  const bool lines = STP && ((g.pad & 0x30000) ? ((g.pad & 0x20000) != 0) : (KCH > 1));
  {
    if (lines) {
      const unsigned short* b0 = wave_rows(blockIdx.x);
#pragma unroll
      for (int i = 0; i < 16; i++) xn[i] = aload16s(b0, patch_off(lane, i & 3, g.ldx, i >> 2));
    } else {
      const unsigned short* p0 = xb + tok_of(blockIdx.x) * g.ldx + 16 * h;
#pragma unroll
      for (int i = 0; i < 16; i++) pload16(xn[i], p0 + piece_off(i));
    }
    P.issued += 16;
  }
  P.fill<0>(); P.fill<1>(); P.fill<2>();
  static_assert(FILL_AHEAD == 3, "prologue fills");
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();                       // the parameter rows in LDS are read (by every wave) before the first slot's barrier

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  unsigned short* pend_ptr = cb;                    // where the deferred results go (meaningful while pend_valid)
  bool pend_valid = false;

  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
#ifdef HFTT_STRIP_STAMPS
    const bool bstamp = !LN && S2DBG(g, 4) && blk == (long)blockIdx.x + gridDim.x && tid == 0;
    unsigned long long* bsb = reinterpret_cast<unsigned long long*>(g.ln_mean) + (long)blockIdx.x * 40;
    if (bstamp) bsb[36] = __builtin_amdgcn_s_memtime();
#endif
    int hb = h;                                       // opaque per iteration: column arithmetic that only depends on (h, tile) is
    asm volatile("" : "+v"(hb));                      // otherwise hoisted out of this loop and kept live in dozens of registers
    const long tok = blk * 128 + wave * 32 + j;
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;          // M % 32 == 0 (host check): a wave is all-valid or all-invalid
    const long tokc = tok_of(blk);
    const long nxt = blk + gridDim.x;
    const bool has_next = nxt < nblk;
    const unsigned short* xrow_next = xb + (has_next ? tok_of(nxt) : tokc) * g.ldx + 16 * hb;
    const long rrow = g.res_mod > 0 ? (long)((unsigned)tokc % (unsigned)g.res_mod) : tokc;
    // The parameter rows in LDS do not change from block to block, so LICM would hoist every read of them out of this loop and keep
    // hundreds of values live across it (spills).  An opaque zero added to the address pins the reads inside the iteration.
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const float* prm_b = prm + zero;
    for (int pass = 0; pass < passes; pass++) {
      f32x16 acc[8];
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {                // accumulators start from the bias
        float b[16];
        lds16f(prm_b + pass * 256 + ot * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) acc[ot][q] = b[q];
      }
      for (int kc = 0; kc < KCH; kc++) {
        if (KCH > 1 || pass == 0) {                   // this step's activations: prefetched during the previous step (or the prologue)
          if (lines) {
#pragma unroll
            for (int pr = 0; pr < 4; pr++) {
              patch_put_lines(patchC, lane, xn[4 * pr], xn[4 * pr + 1], xn[4 * pr + 2], xn[4 * pr + 3]);
              patch_get_rows(patchC, j, hb, xf[4 * pr], xf[4 * pr + 1], xf[4 * pr + 2], xf[4 * pr + 3]);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; i++) xf[i] = xn[i];
          }
        }
#ifdef HFTT_STRIP_STAMPS
        if (bstamp && pass == 0 && kc == 0) bsb[37] = __builtin_amdgcn_s_memtime();
#endif
        const bool last_step = (pass == passes - 1) && (kc == KCH - 1);
        const bool pf_x = (KCH > 1) ? true : last_step;      // (past the last block the address falls back to this block's rows: a harmless re-read)
        const unsigned short* pf_src = (KCH > 1 && !last_step) ? (xb + tokc * g.ldx + (kc + 1 == KCH ? 0 : kc + 1) * 256 + 16 * hb) : xrow_next;
        const unsigned short* pf_base = (KCH > 1 && !last_step) ? (wave_rows(blk) + (kc + 1 == KCH ? 0 : kc + 1) * 256) : wave_rows(has_next ? nxt : blk);      // (STP: wave-uniform)
        const bool pf_res = (KCH == 1) || (kc == KCH - 1);
        const unsigned short* res_src = rb + (has_res ? rrow * g.ldr + pass * 256 + 16 * hb : 0);
        // -DHFTT_STRIP_STAMPS builds only (tools/stamp_linear2.sh), HFTT_STRIP2_DEBUG & 4, !LN: thread 0 of each workgroup's SECOND block stamps the shader clock around the phases of every
        // slot of pass 1 into the (otherwise unused) ln_mean buffer: [workgroup][slot 0..7][4] + [workgroup][8][0..1] for the pass epilogue
#ifdef HFTT_STRIP_STAMPS
        const bool stamp = !LN && S2DBG(g, 4) && pass == (int)((S2PAD(g) >> 8) & 3) && kc == 0 && blk == (long)blockIdx.x + gridDim.x && tid == 0;
        unsigned long long* sb = reinterpret_cast<unsigned long long*>(g.ln_mean) + (long)blockIdx.x * 40;
#define L2STAMP(k) do { if (stamp) sb[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define L2STAMP(k) do { } while (0)
#endif
        static_for<8>([&](auto pt_c) __attribute__((always_inline)) {
          constexpr int pt = decltype(pt_c)::value;
          constexpr int BUF = pt & 3;
          L2STAMP(pt * 4 + 0);
          P.template begin_slot<BUF>();
          L2STAMP(pt * 4 + 1);
          const unsigned char* slot = abase + BUF * SLOT_BYTES;
          // fragment i = u * 8 + tile; the ring refill (four pieces) and the slot's share of the block's memory traffic ride along
          // the slot's share of the block's memory traffic rides BETWEEN the MFMAs: the ring refill, two pieces of the next activations
          // (ONE test for both), two pending result pieces (one test), two residual pieces (no test: HR and the k-chunk are compile-time)
          slot_mfmas_mix(slot, [&](int i, bf16x8 a) __attribute__((always_inline)) { acc[i & 7] = mfma32(a, as_frag(xf[2 * pt + (i >> 3)]), acc[i & 7]); },
                         [&](auto i_c) __attribute__((always_inline)) {
                           constexpr int i = decltype(i_c)::value;
                           if (i == 1) P.template fill_piece<BUF, 0>();
                           if (i == 4) {
                             if (pf_x && !S2DBG(g, 32)) {
                               if (lines) {
                                 int ln = lane;
                                 asm volatile("" : "+v"(ln));      // (offsets formed here, not hoisted)
                                 xn[2 * pt] = aload16s(pf_base, patch_off(ln, (2 * pt) & 3, g.ldx, (2 * pt) >> 2));
                                 xn[2 * pt + 1] = aload16s(pf_base, patch_off(ln, (2 * pt + 1) & 3, g.ldx, (2 * pt + 1) >> 2));
                               } else { pload16(xn[2 * pt], pf_src + piece_off(2 * pt)); pload16(xn[2 * pt + 1], pf_src + piece_off(2 * pt + 1)); }
                             }
                           }
                           if (i == 7) {
                             if (pend_valid && !S2DBG(g, 16)) {
                               if (STP) {
                                 int ln = lane;
                                 asm volatile("" : "+v"(ln));      // (piece offsets formed here: hoisted they would be sixteen live registers)
                                 astore16s(pend_base, patch_off(ln, (2 * pt) & 3, g.ldc, (2 * pt) >> 2), pend[2 * pt]);
                                 astore16s(pend_base, patch_off(ln, (2 * pt + 1) & 3, g.ldc, (2 * pt + 1) >> 2), pend[2 * pt + 1]);
                               } else { astore16(pend_ptr + piece_off(2 * pt), pend[2 * pt]); astore16(pend_ptr + piece_off(2 * pt + 1), pend[2 * pt + 1]); }
                             }
                           }
                           if (i == 10) { if (HR && pf_res) { pload16(pend[2 * pt], res_src + piece_off(2 * pt)); pload16(pend[2 * pt + 1], res_src + piece_off(2 * pt + 1)); } }
                           if (i == 12) P.template fill_close<BUF>();
                         });
          L2STAMP(pt * 4 + 2);
          L2STAMP(pt * 4 + 3);
          if (pt == 7) pend_valid = false;
        });
        L2STAMP(32);
      }
      // ---------------- epilogue of this pass: results into the pending registers ----------------
            const uint64_t rowq = ((uint64_t)tok * (uint64_t)g.N) >> 2;
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
        const int col0 = pass * 256 + ot * 32 + 16 * hb;
        float v[16];
        if (plain) {                                  // (wave-uniform) no ReLU, unit scale: the common projection -- nothing to do per element
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] = acc[ot][q];
        } else {
#pragma unroll
          for (int q = 0; q < 16; q++) {
            float t = acc[ot][q];
            if (!LN && relu) t = fmaxf(t, 0.f);
            v[q] = t * g.out_scale;
          }
        }
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.drop_site, rowq + (col0 >> 2), thr, inv_keep);
        if (has_res) {
          float r[16];
          unpack8(pend[2 * ot], r); unpack8(pend[2 * ot + 1], r + 8);
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] += r[q];
        }
        if (LN) {
#pragma unroll
          for (int q = 0; q < 16; q++) acc[ot][q] = v[q];
        } else if (STP) {
          patch_put(patchC, j, hb, ot & 1, pack8u(v), pack8u(v + 8));
          if (ot & 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) pend[4 * (ot >> 1) + k] = patch_get(patchC, lane, k);
          }
        } else {
          pend[2 * ot] = pack8u(v);
          pend[2 * ot + 1] = pack8u(v + 8);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LN) {
        const unsigned short* pwave = preb + (blk * 128 + wave * 32) * g.ldc;
        ln_rows(acc, prm_b + g.N, prm_b + g.N + 256, hb, g.ln_mean, g.ln_rstd, tok, wave_ok, g.pre_ln_out != nullptr,
                preb + tok * g.ldc + 16 * hb, [&](int n) { P.issued += n; },
                [&](int ot, u4v a, u4v b) __attribute__((always_inline)) {
                  if (STP) {
                    patch_put(patchC, j, hb, ot & 1, a, b);
                    if (ot & 1) {
#pragma unroll
                      for (int k = 0; k < 4; k++) pend[4 * (ot >> 1) + k] = patch_get(patchC, lane, k);
                    }
                  } else { pend[2 * ot] = a; pend[2 * ot + 1] = b; }
                },
                [&](int ot, const u4v& a, const u4v& b) __attribute__((always_inline)) {
                  if (!STP) return false;
                  patch_put(patchP, j, hb, ot & 1, a, b);
                  if (ot & 1) patch_flush(patchP, lane, pwave, g.ldc, ot >> 1);
                  return true;
                });
      }
      pend_ptr = cb + tok * g.ldc + pass * 256 + 16 * hb;
      pend_base = cb + (blk * 128 + wave * 32) * g.ldc + pass * 256;
      pend_valid = wave_ok;
#ifdef HFTT_STRIP_STAMPS
      if (bstamp && pass == passes - 1) bsb[38] = __builtin_amdgcn_s_memtime();
#endif
#ifdef HFTT_STRIP_STAMPS
      if (!LN && S2DBG(g, 4) && pass == (int)((S2PAD(g) >> 8) & 3) && blk == (long)blockIdx.x + gridDim.x && tid == 0)
        (reinterpret_cast<unsigned long long*>(g.ln_mean) + (long)blockIdx.x * 40)[33] = __builtin_amdgcn_s_memtime();
#endif
    }
  }
  if (pend_valid) {                                   // drain: the last pass's results
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (STP) astore16s(pend_base, patch_off(lane, i & 3, g.ldc, i >> 2), pend[i]);
      else astore16(pend_ptr + piece_off(i), pend[i]);
    }
  }
  P.drain();
#ifdef HFTT_STRIP_STAMPS
  if (!LN && S2DBG(g, 4) && tid == 0) (reinterpret_cast<unsigned long long*>(g.ln_mean) + (long)blockIdx.x * 40)[35] = __builtin_amdgcn_s_memtime();
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// fused two-GEMM block, d = 256, p = 32 * PT: mode 0 = FFN forward + residual + LayerNorm, mode 1 = dX half of its backward
// ---------------------------------------------------------------------------------------------------------------------
// One workgroup per CU: one wave per SIMD, 512 registers, the next block's activations prefetched into a second register set.  (A plan at half
// the registers and two workgroups per CU was built and measured in round 5 -- +3.5 % on the inference form, -4 .. -8 % on the training
// forms, compiler-limited -- and removed in round 6: docs/HISTORY.md.)
// STP (round 5): results leave as whole 128-byte lines through wave-private LDS patches (strip_pipe.h: patch_put / patch_flush) instead of
// 16-byte row pieces -- the stored hidden (pairs of tiles, four pieces behind the odd tile's second GEMM), the pre-LayerNorm rows and the output.
template <int MODE, int PT, bool STP>
__global__ __launch_bounds__(256, 1) void strip_mlp2_kernel(const hftt_ffn_desc g) {
  static_assert(PT % 4 == 0, "the gate prefetch ring and the slot buffers assume PT % 4 == 0");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  constexpr int p = PT * 32;
  const long nblk = ((long)g.M + 127) / 128;
  const long my_blocks = (nblk - (long)blockIdx.x + gridDim.x - 1) / gridDim.x;
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // b1[p] | b2[256] | gamma[256] | beta[256]
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(g.x);
  unsigned short* yb = reinterpret_cast<unsigned short*>(g.y);
  unsigned short* preb = reinterpret_cast<unsigned short*>(g.pre_ln_out);
  const unsigned short* rb = reinterpret_cast<const unsigned short*>(g.residual);
  const bool has_res = (MODE == 1) && g.residual != nullptr;
  unsigned char* patchY = smem + RING_BYTES + 4 * (p + 768) + (STP ? wave * 3 * PATCH_BYTES : 0);      // output | pre-LayerNorm | hidden
  unsigned char* patchP = patchY + PATCH_BYTES;
  unsigned char* patchH = patchY + 2 * PATCH_BYTES;

  // Round 5: the constant-wait pipe of the linear kernel (always-fill ring, s_waitcnt vmcnt(8) at every slot boundary) instead of the counted
  // waits of PipeDyn -- ~25 scalar instructions and a branch tree per slot, a quarter of the 530 cycles per slot the kernel's skeleton costs
  // with every mechanism switched off (tools/ablate_mlp2.sh, bits 451).  It used to tip this kernel into scratch; the whole-line store path
  // freed the registers (no 64-bit row pointers per lane).
  Pipe P;
  P.w = g.w; P.S = 2 * PT; P.nofill = S2DBG(g, 1); P.fill_pos = 0; P.nobar = S2DBG(g, 2);
  P.ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  P.wave = wave; P.lane = lane; P.issued = 0;
  (void)my_blocks;

  auto tok_of = [&](long blk) { const long t = blk * 128 + wave * 32 + j; return t < g.M ? t : (long)g.M - 1; };

  for (int i = tid; i < p; i += 256) prm[i] = (MODE == 0 && g.b1 != nullptr) ? g.b1[i] : 0.f;
  prm[p + tid] = (MODE == 0 && g.b2 != nullptr) ? g.b2[tid] : 0.f;
  if (MODE == 0) { prm[p + 256 + tid] = g.ln_gamma[tid]; prm[p + 512 + tid] = g.ln_beta[tid]; }

  // xf: the block's activations; after the last first-GEMM of the block (mode 1) its registers receive the residual rows of the final
  // epilogue.  xn: the next block's activations, one piece per hidden tile.  Results are stored from the epilogue directly: with 16
  // hidden-tile stores already spread over the block, a deferred set on top overflowed the register file (73-95 spills, and a spill
  // of a register with a load in flight is a wrong answer, not a slow one).
  u4v xf[16], xn[16];
  u4v gt[4][2];                                     // mode 1: stored hidden (the gate) of tiles t .. t+2, ring of four
  {
    const long t0 = tok_of(blockIdx.x);
    const unsigned short* p0 = xb + t0 * g.ldx + 16 * h;
    if (STP) {                                      // whole-line pieces from a wave-uniform base (see strip_linear2_kernel)
      const long r0 = (long)blockIdx.x * 128 + wave * 32;
      const unsigned short* b0 = xb + (r0 < (long)g.M ? r0 : (long)g.M - 32) * g.ldx;
#pragma unroll
      for (int i = 0; i < 16; i++) xn[i] = aload16s(b0, patch_off(lane, i & 3, g.ldx, i >> 2));
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) pload16(xn[i], p0 + piece_off(i));
    }
    P.issued += 16;
    if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const unsigned short* gp = g.gate + t0 * g.ldg + t * 32 + 16 * h;
        pload16(gt[t][0], gp); pload16(gt[t][1], gp + 8);
        P.issued += 2;
      }
    }
  }
  P.fill<0>(); P.fill<1>(); P.fill<2>();
  static_assert(FILL_AHEAD == 3, "prologue fills");
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();                       // the parameter rows in LDS are read (by every wave) before the first slot's barrier

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;                                       // (opaque per iteration: see strip_linear2_kernel)
    asm volatile("" : "+v"(hb));
    const long tok = blk * 128 + wave * 32 + j;
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;
    const long tokc = tok_of(blk);
    const long nxt = blk + gridDim.x;
    const bool has_next = nxt < nblk;
    const long tokn = has_next ? tok_of(nxt) : tokc;
    const unsigned short* xrow_next = xb + tokn * g.ldx + 16 * hb;
    const unsigned short* res_src = rb + (has_res ? tokc * g.ldr + 16 * hb : 0);
    const uint64_t rowq_h = ((uint64_t)tok * (uint64_t)p) >> 2;
    int zero = 0;                                     // (keeps the LDS parameter reads inside the iteration: see strip_linear2_kernel)
    asm volatile("" : "+s"(zero));
    const float* prm_b = prm + zero;
    if (STP) {                            // line pieces -> the lane's own row pieces, through the output patch
#pragma unroll
      for (int pr = 0; pr < 4; pr++) {
        patch_put_lines(patchY, lane, xn[4 * pr], xn[4 * pr + 1], xn[4 * pr + 2], xn[4 * pr + 3]);
        patch_get_rows(patchY, j, hb, xf[4 * pr], xf[4 * pr + 1], xf[4 * pr + 2], xf[4 * pr + 3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) xf[i] = xn[i];
    }

    f32x16 yacc[8];
#pragma unroll
    for (int ot = 0; ot < 8; ot++) {
      float b[16];
      lds16f(prm_b + p + ot * 32 + 16 * hb, b);
#pragma unroll
      for (int q = 0; q < 16; q++) yacc[ot][q] = b[q];
    }

    static_for<PT>([&](auto t_c) __attribute__((always_inline)) {
      constexpr int t = decltype(t_c)::value;
      // ---- first GEMM, hidden tile t ----
      constexpr int BA = (2 * t) & 3, BB = (2 * t + 1) & 3;
      const bool stamp = MODE == 0 && S2DBG(g, 4) && (t == 4 || t == 5) && blk == (long)blockIdx.x + gridDim.x && tid == 0;
      unsigned long long* sb = reinterpret_cast<unsigned long long*>(const_cast<unsigned short*>(g.gate)) + (long)blockIdx.x * 16 + (t - 4) * 8;
      if (stamp) sb[0] = __builtin_amdgcn_s_memtime();
      // The hidden tile's bias is read BEFORE the slot's wait + barrier (round 5).  Behind it, its four ds_read_b128 were the first DS reads of
      // the slot's scheduling region and took four of the six places of slot_mfmas' read-ahead group: the first GEMM's fragments then ran ONE
      // read ahead (ISA: `ds_read; s_waitcnt lgkmcnt(1); v_mfma` sixteen times), every MFMA waiting out most of an LDS round trip.
      f32x16 hacc;
      {
        float b[16];
        lds16f(prm_b + t * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) hacc[q] = b[q];
      }
      P.template begin_slot<BA>();
      asm volatile("" : "+v"(P.lane));                // (the ring's source pointers are formed per slot: with the stream position a constant of every
                                                      //  unrolled slot hipcc hoisted all 32 of them -- 64 registers -- out of the block loop and spilled)
      if (stamp) sb[1] = __builtin_amdgcn_s_memtime();
      {
        const unsigned char* slot = abase + BA * SLOT_BYTES;
        const bool in_blk = (t + 2) < PT;             // mode 1: gate of tile t + 2 (of the next block past the end)
        const unsigned short* gp = (MODE == 1) ? g.gate + (in_blk ? tokc : tokn) * g.ldg + ((t + 2) % PT) * 32 + 16 * hb : nullptr;
        if (!S2DBG(g, 256))
        slot_mfmas<6>(slot, [&](int f, bf16x8 a) __attribute__((always_inline)) { hacc = mfma32(a, as_frag(xf[f]), hacc); }, S2DBG(g, 1024));    // fragment f = 2 * pt + u
        static_for<4>([&](auto q_c) __attribute__((always_inline)) {
          constexpr int q = decltype(q_c)::value;
          P.template fill_piece<BA, q>();
          if (q == 0) {
            if (has_next && !(HFTT_MLP2_CT & 4)) {               // one piece of the next block's activations per tile
              if (STP) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const long r0 = nxt * 128 + wave * 32;
                xn[t] = aload16s(xb + (r0 < (long)g.M ? r0 : (long)g.M - 32) * g.ldx, patch_off(ln, t & 3, g.ldx, t >> 2));
              } else pload16(xn[t], xrow_next + piece_off(t));
              P.issued += 1;
            }
          } else if (q < 3) {
            if (MODE == 1 && (in_blk || has_next)) { pload16(gt[(t + 2) & 3][q - 1], gp + 8 * (q - 1)); P.issued += 1; }
          }
        });
        P.template fill_close<BA>();
      }
      if (stamp) sb[2] = __builtin_amdgcn_s_memtime();
      if (t == PT - 1 && has_res) {                   // xf has fed its last MFMA of the block: fetch the residual rows into it
#pragma unroll
        for (int i = 0; i < 16; i++) pload16(xf[i], res_src + piece_off(i));
        P.issued += 16;
      }
      // ---- middle epilogue: the lane's 16 hidden features of tile t become the B operand of the second GEMM ----
      float v[16];
      const int hcol0 = t * 32 + 16 * hb;
      if (S2DBG(g, 128)) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = hacc[q];
      } else if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = fmaxf(hacc[q], 0.f);
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_h, rowq_h + (hcol0 >> 2), thr, inv_keep);
      } else {
        float gv[16];
        unpack8(gt[t & 3][0], gv); unpack8(gt[t & 3][1], gv + 8);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = gv[q] > 0.f ? hacc[q] * g.gate_scale : 0.f;
      }
      u4v hf[2];
      hf[0] = pack8u(v); hf[1] = pack8u(v + 8);
      // ---- second GEMM, K-slice t; the hidden tile's two stores ride behind its first two MFMA groups ----
      if (stamp) sb[3] = __builtin_amdgcn_s_memtime();
      P.template begin_slot<BB>();
      asm volatile("" : "+v"(P.lane));
      if (stamp) sb[4] = __builtin_amdgcn_s_memtime();
      {
        const unsigned char* slot = abase + BB * SLOT_BYTES;
        const bool st_h = g.h_out != nullptr && wave_ok;
        unsigned short* hp = g.h_out + tok * g.ldh + hcol0;
        const unsigned short* hwave = g.h_out + (blk * 128 + wave * 32) * g.ldh;       // (wave-uniform)
        if (STP && st_h) patch_put(patchH, j, hb, t & 1, hf[0], hf[1]);
        if (!S2DBG(g, 256))
        slot_mfmas<6>(slot, [&](int i, bf16x8 a) __attribute__((always_inline)) { yacc[i & 7] = mfma32(a, as_frag(hf[i >> 3]), yacc[i & 7]); }, S2DBG(g, 1024));
        static_for<4>([&](auto q_c) __attribute__((always_inline)) {
          constexpr int q = decltype(q_c)::value;
          P.template fill_piece<BB, q>();
          if (STP) {
            if ((t & 1) && st_h) {
              int ln = lane;
              asm volatile("" : "+v"(ln));            // (the piece offsets are formed here: hoisted out of the block loop they were spilled and reloaded between the MFMAs)
              astore16s(hwave, patch_off(ln, q, g.ldh, t >> 1), patch_get(patchH, ln, q)); P.issued += 1;
            }
          } else if (q < 2 && st_h) { astore16(hp + 8 * q, hf[q]); P.issued += 1; }
        });
        P.template fill_close<BB>();
      }
      if (stamp) sb[5] = __builtin_amdgcn_s_memtime();
    });

    // ---------------- final epilogue of the block ----------------
        const uint64_t rowq = ((uint64_t)tok * 256ull) >> 2;
    unsigned short* yrow = yb + tok * g.ldy + 16 * hb;
    const unsigned short* ywave = yb + (blk * 128 + wave * 32) * g.ldy;               // (wave-uniform)
    const unsigned short* pwave = preb + (blk * 128 + wave * 32) * g.ldy;
    if (S2DBG(g, 64)) {                               // timing switch: no epilogue (the sum of the accumulators keeps them live)
      float sacc = 0.f;
#pragma unroll
      for (int ot = 0; ot < 8; ot++)
#pragma unroll
        for (int q = 0; q < 16; q += 4) sacc += yacc[ot][q];
      if (sacc == 12345.678f) yrow[0] = 1;
      continue;
    }
#pragma unroll
    for (int ot = 0; ot < 8; ot++) {
      const int col0 = ot * 32 + 16 * hb;
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = yacc[ot][q];
      if (MODE == 0 && g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_o, rowq + (col0 >> 2), thr, inv_keep);
      float r[16];
      if (MODE == 0 || has_res) {                     // mode 0: residual = the block input, still in xf; mode 1: just fetched into xf
        unpack8(xf[2 * ot], r); unpack8(xf[2 * ot + 1], r + 8);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] += r[q];
      }
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) yacc[ot][q] = v[q];
      } else if (wave_ok) {
        if (STP) {
          patch_put(patchY, j, hb, ot & 1, pack8u(v), pack8u(v + 8));
          if (ot & 1) patch_flush(patchY, lane, ywave, g.ldy, ot >> 1);
        } else {
          astore16(yrow + piece_off(2 * ot), pack8u(v));
          astore16(yrow + piece_off(2 * ot + 1), pack8u(v + 8));
        }
        P.issued += 2;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 0) {
      ln_rows(yacc, prm_b + p + 256, prm_b + p + 512, hb, g.ln_mean, g.ln_rstd, tok, wave_ok, g.pre_ln_out != nullptr,
              preb + tok * g.ldy + 16 * hb, [&](int n) { P.issued += n; },
              [&](int ot, u4v a, u4v b) __attribute__((always_inline)) {
                if (!wave_ok) return;
                if (STP) {
                  patch_put(patchY, j, hb, ot & 1, a, b);
                  if (ot & 1) patch_flush(patchY, lane, ywave, g.ldy, ot >> 1);
                } else { astore16(yrow + piece_off(2 * ot), a); astore16(yrow + piece_off(2 * ot + 1), b); }
                P.issued += 2;
              },
              [&](int ot, const u4v& a, const u4v& b) __attribute__((always_inline)) {
                if (!STP) return false;
                patch_put(patchP, j, hb, ot & 1, a, b);
                if (ot & 1) patch_flush(patchP, lane, pwave, g.ldy, ot >> 1);
                return true;
              });
    }
  }
  P.drain();
}

int n_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n = prop.multiProcessorCount;
  }
  return n;
}
bool v2_enabled() {
  const char* e = getenv("HFTT_STRIP_V2");
  return !(e && e[0] == '0');
}
template <typename K>
int set_lds(K kernel, int lds, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { hftt_set_error("%s: hipFuncSetAttribute(%d B LDS) failed: %s", what, lds, hipGetErrorString(e)); return 2; }
  return 0;
}
template <bool LN, int PASSES, int KCH, bool HR, bool STP>
int launch_linear2_stp(const hftt_strip_desc& d, hipStream_t st) {
  const int lds = RING_BYTES + 4 * (d.N + 512) + (STP ? 4 * 2 * PATCH_BYTES : 0);
  static int attr = 0;
  if (lds > attr) { if (int rc = set_lds(strip_linear2_kernel<LN, PASSES, KCH, HR, STP>, lds, "strip_linear2")) return rc; attr = lds; }
  const int cus = n_cus();
  if (cus <= 0) { hftt_set_error("strip_linear2: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((strip_linear2_kernel<LN, PASSES, KCH, HR, STP>), dim3((unsigned)(nblk < cus ? nblk : cus)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("strip_linear2");
  return 0;
}
// HFTT_LINEAR2_PATCH=0: the round-2 row-piece stores (A/B switch; default: whole lines through the LDS patch)
template <bool LN, int PASSES, int KCH, bool HR>
int launch_linear2(const hftt_strip_desc& d, hipStream_t st) {
  const char* e = getenv("HFTT_LINEAR2_PATCH");
  if (e && e[0] == '0') return launch_linear2_stp<LN, PASSES, KCH, HR, false>(d, st);
  return launch_linear2_stp<LN, PASSES, KCH, HR, true>(d, st);
}
template <int MODE, bool STP>
int launch_mlp2(const hftt_ffn_desc& d, hipStream_t st) {
  const int lds = RING_BYTES + 4 * (d.p + 768) + (STP ? 4 * 3 * PATCH_BYTES : 0);
  static int attr = 0;
  if (lds > attr) { if (int rc = set_lds(strip_mlp2_kernel<MODE, 16, STP>, lds, "strip_mlp2")) return rc; attr = lds; }
  const int cus = n_cus();
  if (cus <= 0) { hftt_set_error("strip_mlp2: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  const long grid = cus;
  hipLaunchKernelGGL((strip_mlp2_kernel<MODE, 16, STP>), dim3((unsigned)(nblk < grid ? nblk : grid)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("strip_mlp2");
  return 0;
}
}  // namespace

// -1: this shape / storage is not covered by the pipelined form (the caller launches the general kernel); otherwise the launch status
// HFTT_STRIP2_DEBUG (timing experiments, results garbage; -DHFTT_STRIP2_ABLATE builds only): 1 no ring fills, 2 no barriers, 16 no result stores,
// 32 no activation prefetch (linear); fused block: 64 no final epilogue, 128 no middle epilogue, 256 no MFMAs
static int strip2_debug() { const char* e = getenv("HFTT_STRIP2_DEBUG"); return e ? atoi(e) : 0; }

int hftt_strip_linear2_try(const hftt_strip_desc& d0, hipStream_t st) {
  hftt_strip_desc d = d0;
  d.pad = strip2_debug();
  { const char* e = getenv("HFTT_LINEAR2_LINES"); if (e) d.pad |= (e[0] == '0' ? 0x10000 : 0x20000); }
  const uint32_t bf = HFTT_SL_X_BF16 | HFTT_SL_C_BF16;
  if (!v2_enabled() || (d.flags & bf) != bf || d.K % 256 != 0 || d.M % 32 != 0 || d.gate != nullptr) return -1;
  if (d.residual != nullptr && !(d.flags & HFTT_SL_RES_BF16)) return -1;
  const int passes = d.N / 256, kch = d.K / 256;
  const bool hr = d.residual != nullptr;
  // the shapes of the model: QKV (3 passes), cross K/V (2), single projections and the dX forms with K = 256 / 512 / 768; LayerNorm forms
#define HFTT_L2(LN_, P_, K_) return hr ? launch_linear2<LN_, P_, K_, true>(d, st) : launch_linear2<LN_, P_, K_, false>(d, st)
  if (d.ln_gamma != nullptr) {
    if (kch == 1) HFTT_L2(true, 1, 1);
    if (kch == 2) HFTT_L2(true, 1, 2);
    if (kch == 3) HFTT_L2(true, 1, 3);
    return -1;
  }
  if (kch == 1 && passes == 1) HFTT_L2(false, 1, 1);
  if (kch == 1 && passes == 2) HFTT_L2(false, 2, 1);
  if (kch == 1 && passes == 3) HFTT_L2(false, 3, 1);
  if (kch == 2 && passes == 1) HFTT_L2(false, 1, 2);
  if (kch == 3 && passes == 1) HFTT_L2(false, 1, 3);
#undef HFTT_L2
  return -1;
}
int hftt_strip_mlp2_try(const hftt_ffn_desc& d0, hipStream_t st) {
  hftt_ffn_desc d = d0;
  d.pad = strip2_debug();
  if (!v2_enabled() || d.p != 512 || d.M % 32 != 0) return -1;
  if (d.mode == 0 && d.residual != nullptr) return -1;
  // Whole-line stores through the LDS patches (STP), chosen per form by measurement (round 5, S_e, same box, row pieces -> patches): forward
  // with the hidden and the pre-LayerNorm rows saved 288.0 -> 269.8 us (253.7 -> 231.7 without dropout); inference form 191.5 -> 195.1 and
  // the backward 307.2 -> 313.5, so those keep the row pieces.  HFTT_MLP2_PATCH=0 / 1 forces either (A/B, tests).
  const char* e = getenv("HFTT_MLP2_PATCH");
  const bool stp = e ? (e[0] != '0') : (d.mode == 0 && (d.h_out != nullptr || d.pre_ln_out != nullptr));
  if (!stp) return d.mode == 0 ? launch_mlp2<0, false>(d, st) : launch_mlp2<1, false>(d, st);
  return d.mode == 0 ? launch_mlp2<0, true>(d, st) : launch_mlp2<1, true>(d, st);
}
