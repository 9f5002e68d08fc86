// Strip linear kernel, fourth form ("v4"): waves that multiply and waves that move data.
//
// Why.  In the second form (strip_gemm2.hip) one wave per SIMD does everything, and a wave issues in order: while it waits for a slot of
// the vector-memory pipe's queue (a result store, an activation load) its MFMAs do not issue either.  Measured on the QKV projection:
// 188 us = 117 us without the stores + 70 us, i.e. matrix work and HBM traffic in series, although per ring slot the memory pipe is busy
// ~1,100-1,350 cycles against 512 cycles of MFMAs -- the kernel should be bound by its traffic alone.
//
// Here a workgroup has eight waves, two per SIMD, 256 registers each.  Waves 0-3 ("compute") own a 32-token strip each exactly as before,
// but touch no global memory inside the loop: weight fragments AND their strip's activations come from LDS, results go to an LDS staging
// ring.  Waves 4-7 ("movers", wave 4 + w serves strip w with the SAME lane -> (token, column) map) issue every vector-memory instruction:
// the weight ring's LDS-DMA fills, the activations of the next step (LDS-DMA into the strip's 16 KB area, each pair of 16-byte pieces
// refreshed one slot after the compute wave consumed it -- one buffer, no second copy), and the result stores (two staged pieces per slot,
// read back from LDS one slot after they were written).  The SIMD's issue arbiter interleaves the two waves, so a store waiting for the
// memory pipe no longer holds an MFMA back.
//
// Synchronisation is ONE workgroup barrier per ring slot, executed by all eight waves from loop nests of the same shape (one barrier per
// slot in each role, plus the prologue's and the tail's), so both roles execute the same number of barriers by construction; there are
// no flags and no spin waits.  Covered: bf16 tensors, no LayerNorm, no residual, K / 256 in {1, 2, 3}, N / 256 in {1, 2, 3}.
//
// RESULT (MI355X; tools/chk_forms.py, tools/ablate_v4.py).  Bit-identical to the second form on every shape.  Faster where the launch is short
// (M = 38,432: 5-10 % on all five shapes) and on K = N = 256 at S_e (78 against 85 us); NOT faster on the big shapes (QKV at S_e 176-200 us
// against 179 us) -- so it stays OPT-IN (HFTT_STRIP_V4=1).  What the role split makes measurable (QKV, S_e, HFTT_STRIP4_DEBUG bits):
//     compute waves alone (movers idle, barriers kept)      107 us   (MFMAs + LDS reads only, no barriers: 95 us = 1,170 cycles per slot of 16 MFMAs)
//     mover waves alone (compute idle)                      158 us   = ring fills alone 57-66 us + result stores alone 85 us + activations ~38 us, nearly additive
//     movers without the ring fills                         132 us
// i.e. the kernel is bound by what one CU's vector-memory path moves per slot, and 60 % of those bytes are WEIGHTS: every 128-token block
// pulls the whole weight stream L2 -> LDS again (805 MB per QKV launch at ~12 TB/s chip-wide, against 537 MB of HBM traffic at ~4 TB/s).
// Deeper queues do not help (an 8-slot ring with vmcnt(24) in the second form: QKV 183 us, K = N = 256 78 us), nor does giving the HBM traffic
// to waves that never wait for the ring.  The lever that is left is fewer weight bytes per token: a wider token tile per weight fetch
// (64 tokens per wave halves both the L2 -> LDS stream and the LDS -> register reads per MFMA), which needs the accumulators of two strips
// (256 registers) -- a different register plan, not a different schedule.  DESIGN.md section 4.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../hftt_common.h"
#include "../hftt_host.h"
#include "../strip_internal.h"
#include "../../../include/hftt_hip.h"

namespace {

#include "../strip_pipe.h"

constexpr int XBUF_BYTES = 65536;                   // 4 strips x 16 pieces x (64 lanes x 16 B)
constexpr int STG_BYTES = 16384;                    // 4 strips x 2 entries x 2 pieces x 1 KB

// two LDS-DMA pieces: 16 B per lane from gsrc and from gsrc + 16 B, to lds_dst (+ 16 * lane) and lds_dst + 1 KB.  The instruction offset
// advances the global AND the LDS address, so the second piece runs with M0 = lds_dst + 1024 - 16 and offset 16.
__device__ __forceinline__ void glds16_pair(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  const unsigned second = lds_dst + 1008u;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off offset:16\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst), "s"(second));
}

template <int PASSES, int KCH>
__global__ __launch_bounds__(512, 1) void strip_linear4_kernel(const hftt_strip_desc g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mover = wave8 >= 4;
  const int wave = wave8 & 3;                       // the strip this wave computes (0-3) or serves (4-7)
  const int j = lane & 31, h = lane >> 5;
  const long nblk = ((long)g.M + 127) / 128;
  unsigned char* xbuf = smem + RING_BYTES + wave * 16384;          // this strip's activations: piece i at i * 1 KB + lane * 16
  unsigned char* stg = smem + RING_BYTES + XBUF_BYTES + wave * 4096;   // this strip's staging: entry e, piece q at e * 2 KB + q * 1 KB + lane * 16
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES + XBUF_BYTES + STG_BYTES);      // bias[N]
  const unsigned lds0 = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  const unsigned x_lds = lds0 + RING_BYTES + (unsigned)wave * 16384u;
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(g.x);
  unsigned short* cb = reinterpret_cast<unsigned short*>(g.C);
  const bool relu = g.flags & HFTT_SL_RELU;
  const bool plain = !relu && g.out_scale == 1.0f;
  constexpr int S = PASSES * KCH * 8;               // ring slots per block
  const unsigned dbg = g.pad;                       // HFTT_STRIP4_DEBUG (timing experiments, results garbage): 1 no stores, 2 no activation refresh, 4 no ring fills, 8 no staging, 16 no barriers, 64 compute waves idle

  auto tok_of = [&](long blk) { const long t = blk * 128 + wave * 32 + j; return t < g.M ? t : (long)g.M - 1; };
  auto wsrc = [&](int pos) { return g.w + ((long)pos * 16 + wave * 4) * 512 + lane * 8; };      // mover w moves fragments 4w .. 4w+3 of a slot

  // ---- prologue: bias to LDS; movers fetch the first block's activations and the first three ring slots
  for (int i = tid; i < g.N; i += 512) prm[i] = g.bias != nullptr ? g.bias[i] : 0.f;
  int issued = 0, fill_pos = 0;
  int mark[NSLOT] = {0, 0, 0, 0};
  if (mover) {
    const unsigned short* p0 = xb + tok_of(blockIdx.x) * g.ldx + 16 * h;
#pragma unroll
    for (int p = 0; p < 8; p++) glds16_pair(p0 + p * 32, x_lds + (unsigned)p * 2048u);
#pragma unroll
    for (int b = 0; b < FILL_AHEAD; b++) {
      glds16x4(wsrc(fill_pos), lds0 + (unsigned)b * SLOT_BYTES + (unsigned)wave * 4096u);
      fill_pos = (fill_pos + 1 == S) ? 0 : fill_pos + 1;
    }
    HFTT_WAITVM(0);
  }
  static_assert(FILL_AHEAD == 3 && NSLOT == 4, "ring geometry");
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  if (mover) {
    // =====================================================================================================================
    // mover wave: every vector-memory instruction of strip `wave`.  Same loop nest and the same barriers as the compute waves.
    // (Tried: waves 4-5 only filling the ring, waves 6-7 only HBM traffic for two strips each, so that the ring wait no longer
    // covers HBM operations -- slower, 196 us against 176 us on QKV: two waves drive the HBM traffic less well than four.)
    // =====================================================================================================================
    unsigned short* pend_ptr = cb;                  // where the compute wave's pending results go (this lane's row view)
    bool pend_valid = false;
    unsigned short* prev_ptr = cb;                  // the same, one slot ago (the stores run one slot behind the staging writes)
    bool prev_valid = false;
    const unsigned short* carry_src = xb;           // the refresh of pieces 14 / 15 happens in slot 0 of the NEXT step
    bool carry_do = false;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
      const long tok = blk * 128 + wave * 32 + j;
      const bool wave_ok = (blk * 128 + wave * 32) < g.M;
      const long tokc = tok_of(blk);
      const long nxt = blk + gridDim.x;
      const bool has_next = nxt < nblk;
      for (int pass = 0; pass < PASSES; pass++) {
        for (int kc = 0; kc < KCH; kc++) {
          const bool last_step = (pass == PASSES - 1) && (kc == KCH - 1);
          // what the NEXT step multiplies: another 256-column chunk of this block's rows, or the first chunk of the next block's
          const bool do_refresh = (KCH > 1) ? (!last_step || has_next) : (last_step && has_next);
          const unsigned short* nx_src = last_step ? (xb + (has_next ? tok_of(nxt) : tokc) * g.ldx + 16 * h)
                                                   : (xb + tokc * g.ldx + (kc + 1 == KCH ? 0 : kc + 1) * 256 + 16 * h);
          static_for<8>([&](auto pt_c) __attribute__((always_inline)) {
            constexpr int pt = decltype(pt_c)::value;
            constexpr int BUF = pt & 3;
            constexpr int pp = (pt + 7) & 7;         // the slot before this one
            wait_vmcnt_dyn(issued - mark[BUF]);      // this slot's weights have landed (and everything issued before them)
            if (!(dbg & 16)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (prev_valid && !(dbg & 1)) {          // results staged during the previous slot -> global memory
              const u4v a = *reinterpret_cast<const u4v*>(stg + (pp & 1) * 2048 + lane * 16);
              const u4v b = *reinterpret_cast<const u4v*>(stg + (pp & 1) * 2048 + 1024 + lane * 16);
              astore16(prev_ptr + piece_off(2 * pp), a);
              astore16(prev_ptr + piece_off(2 * pp + 1), b);
              issued += 2;
            }
            // the activation pieces consumed during the previous slot -> what their next use needs
            if (pt == 0) {
              if (carry_do && !(dbg & 2)) { glds16_pair(carry_src + pp * 32, x_lds + (unsigned)pp * 2048u); issued += 2; }
            } else {
              if (do_refresh && !(dbg & 2)) { glds16_pair(nx_src + pp * 32, x_lds + (unsigned)pp * 2048u); issued += 2; }
            }
            // the weight ring, three slots ahead (past the end the stream wraps: harmless extra slots, drained before the kernel ends)
            if (!(dbg & 4)) {
              glds16x4(wsrc(fill_pos), lds0 + (unsigned)((BUF + FILL_AHEAD) & (NSLOT - 1)) * SLOT_BYTES + (unsigned)wave * 4096u);
              issued += 4;
            }
            mark[(BUF + FILL_AHEAD) & (NSLOT - 1)] = issued;
            fill_pos = (fill_pos + 1 == S) ? 0 : fill_pos + 1;
            prev_ptr = pend_ptr; prev_valid = pend_valid;
            if (pt == 7) pend_valid = false;
          });
          carry_do = do_refresh; carry_src = nx_src;
        }
        pend_ptr = cb + tok * g.ldc + pass * 256 + 16 * h;
        pend_valid = wave_ok;
      }
    }
    // tail: the entry staged in the very last slot
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (prev_valid) {
      const u4v a = *reinterpret_cast<const u4v*>(stg + 2048 + lane * 16);
      const u4v b = *reinterpret_cast<const u4v*>(stg + 2048 + 1024 + lane * 16);
      astore16(prev_ptr + piece_off(14), a);
      astore16(prev_ptr + piece_off(15), b);
    }
    HFTT_WAITVM(0);                                   // nothing may still be on its way into LDS when the workgroup ends
    return;
  }

  // =======================================================================================================================
  // compute wave: weight fragments and activations from LDS, results to the staging ring; no global memory inside the loop
  // =======================================================================================================================
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  u4v pend[16];                                     // the finished pass's results, two pieces per slot into the staging ring
  unsigned short* pend_ptr = cb;
  bool pend_valid = false;
  u4v xc0 = *reinterpret_cast<const u4v*>(xbuf + lane * 16), xc1 = *reinterpret_cast<const u4v*>(xbuf + 1024 + lane * 16);      // this slot's two activation pieces
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;
    asm volatile("" : "+v"(hb));
    const long tok = blk * 128 + wave * 32 + j;
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const float* prm_b = prm + zero;
    for (int pass = 0; pass < PASSES; pass++) {
      f32x16 acc[8];
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
        float b[16];
        lds16f(prm_b + pass * 256 + ot * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) acc[ot][q] = b[q];
      }
      for (int kc = 0; kc < KCH; kc++) {
        static_for<8>([&](auto pt_c) __attribute__((always_inline)) {
          constexpr int pt = decltype(pt_c)::value;
          constexpr int BUF = pt & 3;
          wait_lgkm0();                              // the staging entry of the last slot is written
          if (!(dbg & 16)) __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          const unsigned char* slot = abase + BUF * SLOT_BYTES;
          u4v xn0 = xc0, xn1 = xc1;
          if (!(dbg & 64))
          slot_mfmas_mix(slot, [&](int i, bf16x8 a) __attribute__((always_inline)) { acc[i & 7] = mfma32(a, as_frag((i >> 3) ? xc1 : xc0), acc[i & 7]); },
                         [&](auto i_c) __attribute__((always_inline)) {
                           constexpr int i = decltype(i_c)::value;
                           if (i == 3) {              // the next slot's pieces (slot 7: pieces 0 / 1 of the next step, refreshed six slots ago)
                             constexpr int np = (pt + 1) & 7;
                             xn0 = *reinterpret_cast<const u4v*>(xbuf + (2 * np) * 1024 + lane * 16);
                             xn1 = *reinterpret_cast<const u4v*>(xbuf + (2 * np + 1) * 1024 + lane * 16);
                           }
                           if (i == 9) {
                             if (pend_valid && !(dbg & 8)) {
                               *reinterpret_cast<u4v*>(stg + (pt & 1) * 2048 + lane * 16) = pend[2 * pt];
                               *reinterpret_cast<u4v*>(stg + (pt & 1) * 2048 + 1024 + lane * 16) = pend[2 * pt + 1];
                             }
                           }
                         });
          xc0 = xn0; xc1 = xn1;
          if (pt == 7) pend_valid = false;
        });
      }
      // ---------------- epilogue of this pass: results into the pending registers ----------------
      const uint64_t rowq = ((uint64_t)tok * (uint64_t)g.N) >> 2;
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
        const int col0 = pass * 256 + ot * 32 + 16 * hb;
        float v[16];
        if (plain) {
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] = acc[ot][q];
        } else {
#pragma unroll
          for (int q = 0; q < 16; q++) {
            float t = acc[ot][q];
            if (relu) t = fmaxf(t, 0.f);
            v[q] = t * g.out_scale;
          }
        }
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.drop_site, rowq + (col0 >> 2), thr, inv_keep);
        pend[2 * ot] = pack8u(v);
        pend[2 * ot + 1] = pack8u(v + 8);
        __builtin_amdgcn_sched_barrier(0);
      }
      pend_ptr = cb + tok * g.ldc + pass * 256 + 16 * hb;
      pend_valid = wave_ok;
    }
  }
  // ---- tail: the movers store the entry staged in the very last slot; the last pass's results are stored from here
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (pend_valid) {
#pragma unroll
    for (int i = 0; i < 16; i++) astore16(pend_ptr + piece_off(i), pend[i]);
  }
}

int n_cus4() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n = prop.multiProcessorCount;
  }
  return n;
}

template <int PASSES, int KCH>
int launch_linear4(const hftt_strip_desc& d, hipStream_t st) {
  const int lds = RING_BYTES + XBUF_BYTES + STG_BYTES + 4 * d.N;
  static int attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(strip_linear4_kernel<PASSES, KCH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { hftt_set_error("strip_linear4: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e)); return 2; }
    attr = lds;
  }
  const int cus = n_cus4();
  if (cus <= 0) { hftt_set_error("strip_linear4: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((strip_linear4_kernel<PASSES, KCH>), dim3((unsigned)(nblk < cus ? nblk : cus)), dim3(512), lds, st, d);
  HFTT_CHECK_LAUNCH("strip_linear4");
  return 0;
}

}  // namespace

// -1: not covered (the caller tries the other forms); otherwise the launch status
int hftt_strip_linear4_try(const hftt_strip_desc& d0, hipStream_t st) {
  hftt_strip_desc d = d0;
  { const char* e = getenv("HFTT_STRIP4_DEBUG"); d.pad = e ? atoi(e) : 0; }
  const char* e = getenv("HFTT_STRIP_V4");          // (read at every call: the tests switch forms inside one process)
  const bool enabled = e && e[0] == '1';
  const uint32_t bf = HFTT_SL_X_BF16 | HFTT_SL_C_BF16;
  if (!enabled || (d.flags & bf) != bf || d.K % 256 != 0 || d.N % 256 != 0 || d.M % 32 != 0 || d.gate != nullptr) return -1;
  if (d.residual != nullptr || d.ln_gamma != nullptr) return -1;
  if (d.ldx % 8 != 0 || ((uintptr_t)d.x & 15) != 0) return -1;        // 16-byte LDS-DMA pieces
  const int passes = d.N / 256, kch = d.K / 256;
  if (kch == 1 && passes == 1) return launch_linear4<1, 1>(d, st);
  if (kch == 1 && passes == 2) return launch_linear4<2, 1>(d, st);
  if (kch == 1 && passes == 3) return launch_linear4<3, 1>(d, st);
  if (kch == 2 && passes == 1) return launch_linear4<1, 2>(d, st);
  if (kch == 3 && passes == 1) return launch_linear4<1, 3>(d, st);
  return -1;
}
