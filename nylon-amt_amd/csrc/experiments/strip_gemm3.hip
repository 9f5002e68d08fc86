// Strip linear kernel, third form: the ACTIVATIONS also come through LDS, and the four waves split the memory roles.
//
// Why.  vmcnt retires in issue order, per wave.  In the second form (strip_gemm2.hip) every wave both streams weight fragments from
// L2 (LDS-DMA, a few hundred cycles) and prefetches the next activations from HBM into registers (2-3 us under load): the wait for a
// ring slot then also waits for every older HBM load of that wave.  Stamps (tools/stamp_linear2.sh) show it: a pass that carries the
// prefetch takes 23k ticks against 8k for the passes without it -- a third of the kernel.
// Here waves 0 and 1 ("fillers") issue ALL weight fragments of a slot (8 pieces each) and are the only ones that wait on vmcnt per
// slot; waves 2 and 3 ("fetchers") bring the activations of ALL four waves by LDS-DMA into a 64 KB activation buffer, two whole token rows
// per instruction, and wait for them once per step.  Every wave still does a quarter of the MFMAs, stores its own results and keeps its strip in
// registers.  The activation buffer is read at the top of a step's first slot and refilled over the following slots.  The second prefetch
// register set of the second form is gone.
//
// RESULT (MI355X, S_e): not faster -- QKV 185 us (scattered 16-byte pieces in a burst), 247 us (whole rows in a burst), 206 us (whole rows
// spread over the block) against 186 us for the second form; the pass that carries the fetch stays 2-2.6x the others however the bytes are
// asked for.  What the experiments of this file and HFTT_STRIP2_DEBUG=16 / 32 in strip_gemm2.hip say: without result stores QKV takes 117 us,
// without activation loads 141 us, with neither 118 us -- the CU's single vector-memory pipe (~10 B/clk) and the matrix work run one after the
// other rather than side by side, whichever wave issues the memory instructions.  OPT-IN (HFTT_STRIP_V3=1); kept for the next attempt.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../hftt_common.h"
#include "../hftt_host.h"
#include "../strip_internal.h"
#include "../../../include/hftt_hip.h"

namespace {

#include "../strip_pipe.h"

constexpr int XBUF_BYTES = 128 * 512;                // [token][256 features] bf16, rows rotated: one 256-feature chunk of a 128-token block

// one LDS-DMA piece (16 B per lane): per-lane global source, LDS destination lds_dst + 16 * lane
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst));
}

template <bool LN, int PASSES, int KCH, bool HR>
__global__ __launch_bounds__(256, 1) void strip_linear3_kernel(const hftt_strip_desc g) {
  static_assert(PASSES == 1 || KCH == 1, "shapes of the model: several output passes OR several k-chunks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const bool filler = wave < 2;                       // (wave-uniform) waves 0, 1 stream the weights; waves 2, 3 fetch the activations
  constexpr int passes = PASSES;
  constexpr int S = passes * KCH * 8;                 // ring slots per block
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES + XBUF_BYTES);      // bias[N] | gamma[256] | beta[256]
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(g.x);
  unsigned short* cb = reinterpret_cast<unsigned short*>(g.C);
  unsigned short* preb = reinterpret_cast<unsigned short*>(g.pre_ln_out);
  const unsigned short* rb = reinterpret_cast<const unsigned short*>(g.residual);
  const bool relu = g.flags & HFTT_SL_RELU;
  const bool plain = !relu && g.out_scale == 1.0f;
  const unsigned ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  const unsigned xbuf = ring + RING_BYTES;
  int fill_pos = 0;                                   // stream position (0 .. S-1) of the next slot to fetch (wraps: see strip_gemm2.hip, Pipe)

  auto tok_of = [&](long blk, int w) { const long t = blk * 128 + w * 32 + j; return t < g.M ? t : (long)g.M - 1; };
  // filler wave w: fragments 8w .. 8w+7 of the next stream slot -> ring buffer buf
  auto fill = [&](int buf) __attribute__((always_inline)) {
    const unsigned short* src = g.w + ((long)fill_pos * 16 + wave * 8) * 512 + lane * 8;
    const unsigned dst = ring + (unsigned)buf * SLOT_BYTES + (unsigned)wave * 8192u;
    glds16x4(src, dst);
    glds16x4(src + 4 * 512, dst + 4096u);
  };
  auto advance = [&]() __attribute__((always_inline)) { fill_pos = (fill_pos + 1 == S) ? 0 : fill_pos + 1; };
  // fetcher wave f = wave - 2, instruction k = 0 .. 31 of its half: the TWO whole 512-byte rows 2q, 2q+1 (q = 32f + k) of (block blk, chunk):
  // 8 full cache lines per instruction.  (A first version fetched, per instruction, the 16 bytes each lane of a reader wave needs: 32
  // different lines touched for a quarter of their bytes each -- the pass that carried those fetches ran 2.6x slower than the others.)
  // The buffer is row-major, 512 B per token, and row r is stored ROTATED by r chunks of 16 B (the lane that lands at position p fetches chunk
  // (p - r) mod 32), so that the readers' "same chunk of 32 consecutive rows" hits 32 different positions instead of one bank group.
  auto xfetch = [&](long blk, int chunk, int k) __attribute__((always_inline)) {
    const int q = 32 * (wave - 2) + k;
    const int r = 2 * q + h;                          // (lane >> 5 = h: which of the two rows; j = the position)
    const long t = blk * 128 + r;
    const long tc = t < g.M ? t : (long)g.M - 1;
    glds16(xb + tc * g.ldx + chunk * 256 + ((j - r) & 31) * 8, xbuf + (unsigned)q * 1024u);
  };

  // ---- prologue: parameters to LDS, the first block's first chunk, the first three ring slots
  for (int i = tid; i < g.N; i += 256) prm[i] = g.bias != nullptr ? g.bias[i] : 0.f;
  if (LN) { prm[g.N + tid] = g.ln_gamma[tid]; prm[g.N + 256 + tid] = g.ln_beta[tid]; }
  if (filler) {
    fill(0); advance(); fill(1); advance(); fill(2); advance();
  } else {
#pragma unroll
    for (int k = 0; k < 32; k++) xfetch(blockIdx.x, 0, k);
    advance(); advance(); advance();
    HFTT_WAITVM(0);                                   // the activations have landed before the barrier below
  }
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();

  u4v xf[16], pend[16];
  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  const unsigned char* xrow = smem + RING_BYTES + (wave * 32 + j) * 512;       // this lane's token row in the activation buffer
  unsigned short* pend_ptr = cb;                    // where the deferred results go (meaningful while pend_valid)
  bool pend_valid = false;
  // the activation fetch is SPREAD: NX instructions per fetcher and slot, so that it ends about three slots before the buffer is read again
  // (a burst of 8 per slot cost the whole workgroup ~1,700 ticks per slot: LDS-DMA issue stalls while the CU's miss queue is full)
  constexpr int XSLOTS = (KCH > 1 ? 8 : PASSES * 8) - 1 - 3;
  constexpr int NX = (32 + XSLOTS - 1) / XSLOTS < 1 ? 1 : (32 + XSLOTS - 1) / XSLOTS;
  int xk = 32, xsrc_chunk = 0;
  long xsrc_blk = 0;

  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;                                       // opaque per iteration (see strip_gemm2.hip)
    asm volatile("" : "+v"(hb));
    const long tok = blk * 128 + wave * 32 + j;
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;          // M % 32 == 0 (host check): a wave is all-valid or all-invalid
    const long tokc = tok_of(blk, wave);
    const long nxt = blk + gridDim.x;
    const long nxt_c = nxt < nblk ? nxt : blk;        // past the last block: a harmless re-fetch of this block
    const long rrow = g.res_mod > 0 ? (long)((unsigned)tokc % (unsigned)g.res_mod) : tokc;
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
        const bool need_x = (KCH > 1) || (pass == 0);             // this step reads the activation buffer (and then refills it)
        const long src_blk = (kc + 1 < KCH) ? blk : nxt_c;        // what the buffer is refilled with: the next chunk, or the next block's first
        const int src_chunk = (kc + 1 < KCH) ? kc + 1 : 0;
        const bool pf_res = HR && (kc == KCH - 1);
        const unsigned short* res_src = rb + (HR ? rrow * g.ldr + pass * 256 + 16 * hb : 0);
#ifdef HFTT_STRIP_STAMPS
        // HFTT_STRIP3_STAMP=pass (dev builds): lane 0 of wave 0 (a filler) and of wave 2 (a fetcher) of each workgroup's SECOND block stamp the
        // shader clock around the phases of every slot of that pass into the ln_mean buffer: [wave 0 | wave 2][workgroup][40]
        const bool stamp = !LN && (g.pad & 4) && pass == (int)((g.pad >> 8) & 3) && kc == 0 && blk == (long)blockIdx.x + gridDim.x && (tid == 0 || tid == 128);
        unsigned long long* sb = reinterpret_cast<unsigned long long*>(g.ln_mean) + ((long)(tid >> 7) * gridDim.x + blockIdx.x) * 40;
#define L3STAMP(k) do { if (stamp) sb[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define L3STAMP(k) do { } while (0)
#endif
        static_for<8>([&](auto pt_c) __attribute__((always_inline)) {
          constexpr int pt = decltype(pt_c)::value;
          constexpr int BUF = pt & 3;
          L3STAMP(pt * 4 + 0);
          // top of the slot: fillers wait for the slot's fragments (16 of their DMA pieces were issued after them: constant wait);
          // fetchers wait for the activations once, before the barrier of the slot that reads them
          if (filler) HFTT_WAITVM(16);
          else if (pt == 0 && need_x) HFTT_WAITVM(0);
          __builtin_amdgcn_s_barrier();
          L3STAMP(pt * 4 + 1);
          if (pt == 0 && need_x) {
            xk = 0; xsrc_blk = src_blk; xsrc_chunk = src_chunk;      // the refill that follows this read
#pragma unroll
            for (int i = 0; i < 16; i++) xf[i] = *reinterpret_cast<const u4v*>(xrow + ((((i >> 1) * 4 + 2 * h + (i & 1)) + j) & 31) * 16);   // piece i = chunk 4(i>>1) + 2h + (i&1), rotated by the row
          }
          const unsigned char* slot = abase + BUF * SLOT_BYTES;
          slot_mfmas_mix(slot, [&](int i, bf16x8 a) __attribute__((always_inline)) { acc[i & 7] = mfma32(a, as_frag(xf[2 * pt + (i >> 3)]), acc[i & 7]); },
                         [&](auto i_c) __attribute__((always_inline)) {
                           constexpr int i = decltype(i_c)::value;
                           if (i == 1) {
                             if (filler) fill((BUF + FILL_AHEAD) & (NSLOT - 1));
                             else if (!(need_x && pt == 0) && xk < 32) {        // NX instructions per slot from the slot after the buffer was read
#pragma unroll
                               for (int q = 0; q < NX; q++) xfetch(xsrc_blk, xsrc_chunk, xk + q);
                               xk += NX;
                             }
                           }
                           if (i == 7) { if (pend_valid) { astore16(pend_ptr + piece_off(2 * pt), pend[2 * pt]); astore16(pend_ptr + piece_off(2 * pt + 1), pend[2 * pt + 1]); } }
                           if (i == 10) { if (pf_res) { pload16(pend[2 * pt], res_src + piece_off(2 * pt)); pload16(pend[2 * pt + 1], res_src + piece_off(2 * pt + 1)); } }
                           if (i == 12) advance();
                         });
          L3STAMP(pt * 4 + 2);
          if (pt == 7) pend_valid = false;
        });
        L3STAMP(32);
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
            if (!LN && relu) t = fmaxf(t, 0.f);
            v[q] = t * g.out_scale;
          }
        }
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.drop_site, rowq + (col0 >> 2), thr, inv_keep);
        if (HR) {
          float r[16];
          unpack8(pend[2 * ot], r); unpack8(pend[2 * ot + 1], r + 8);
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] += r[q];
        }
        if (LN) {
#pragma unroll
          for (int q = 0; q < 16; q++) acc[ot][q] = v[q];
        } else {
          pend[2 * ot] = pack8u(v);
          pend[2 * ot + 1] = pack8u(v + 8);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LN) {
        ln_rows(acc, prm_b + g.N, prm_b + g.N + 256, hb, g.ln_mean, g.ln_rstd, tok, wave_ok, g.pre_ln_out != nullptr,
                preb + tok * g.ldc + 16 * hb, [&](int) {},
                [&](int ot, u4v a, u4v b) __attribute__((always_inline)) { pend[2 * ot] = a; pend[2 * ot + 1] = b; });
      }
      pend_ptr = cb + tok * g.ldc + pass * 256 + 16 * hb;
      pend_valid = wave_ok;
#ifdef HFTT_STRIP_STAMPS
      if (!LN && (g.pad & 4) && pass == (int)((g.pad >> 8) & 3) && blk == (long)blockIdx.x + gridDim.x && (tid == 0 || tid == 128))
        (reinterpret_cast<unsigned long long*>(g.ln_mean) + ((long)(tid >> 7) * gridDim.x + blockIdx.x) * 40)[33] = __builtin_amdgcn_s_memtime();
#endif
    }
  }
  if (pend_valid) {                                   // drain: the last pass's results
#pragma unroll
    for (int i = 0; i < 16; i++) astore16(pend_ptr + piece_off(i), pend[i]);
  }
  HFTT_WAITVM(0);                                     // nothing may still be on its way into LDS when the workgroup ends (slots / chunks fetched past the end)
}

int n_cus3() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n = prop.multiProcessorCount;
  }
  return n;
}

template <bool LN, int PASSES, int KCH, bool HR>
int launch_linear3(const hftt_strip_desc& d, hipStream_t st) {
  const int lds = RING_BYTES + XBUF_BYTES + 4 * (d.N + 512);
  static int attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(strip_linear3_kernel<LN, PASSES, KCH, HR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { hftt_set_error("strip_linear3: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e)); return 2; }
    attr = lds;
  }
  const int cus = n_cus3();
  if (cus <= 0) { hftt_set_error("strip_linear3: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((strip_linear3_kernel<LN, PASSES, KCH, HR>), dim3((unsigned)(nblk < cus ? nblk : cus)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("strip_linear3");
  return 0;
}

}  // namespace

// -1: not covered (the caller tries the second form, then the general kernel); otherwise the launch status
int hftt_strip_linear3_try(const hftt_strip_desc& d0, hipStream_t st) {
  hftt_strip_desc d = d0;
  d.pad = 0;
#ifdef HFTT_STRIP_STAMPS
  if (const char* e = getenv("HFTT_STRIP2_DEBUG")) d.pad = atoi(e);
#endif
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("HFTT_STRIP_V3"); enabled = (e && e[0] == '1'); }      // opt-in: measured no faster than the second form (header)
  const uint32_t bf = HFTT_SL_X_BF16 | HFTT_SL_C_BF16;
  if (!enabled || (d.flags & bf) != bf || d.K % 256 != 0 || d.M % 32 != 0 || d.gate != nullptr) return -1;
  if (d.residual != nullptr && !(d.flags & HFTT_SL_RES_BF16)) return -1;
  if (d.ldx % 8 != 0 || ((uintptr_t)d.x & 15) != 0) return -1;        // 16-byte LDS-DMA pieces
  const int passes = d.N / 256, kch = d.K / 256;
  const bool hr = d.residual != nullptr;
#define HFTT_L3(LN_, P_, K_) return hr ? launch_linear3<LN_, P_, K_, true>(d, st) : launch_linear3<LN_, P_, K_, false>(d, st)
  if (d.ln_gamma != nullptr) {
    if (kch == 1) HFTT_L3(true, 1, 1);
    if (kch == 2) HFTT_L3(true, 1, 2);
    if (kch == 3) HFTT_L3(true, 1, 3);
    return -1;
  }
  if (kch == 1 && passes == 1) HFTT_L3(false, 1, 1);
  if (kch == 1 && passes == 2) HFTT_L3(false, 2, 1);
  if (kch == 1 && passes == 3) HFTT_L3(false, 3, 1);
  if (kch == 2 && passes == 1) HFTT_L3(false, 1, 2);
  if (kch == 3 && passes == 1) HFTT_L3(false, 1, 3);
#undef HFTT_L3
  return -1;
}
