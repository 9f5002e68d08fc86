// Strip kernels of the split-operand ("x3") precision mode (gfx950): the nn.Linear layers whose width is a multiple of 256 and the fused
// position-wise feed-forward block, fp32 tensors in HBM, every product in three MFMA passes on 16-bit halves (x3_common.h).
// Contract: include/hftt_hip.h (hftt_strip_linear / hftt_ffn_res_ln_fwd / hftt_ffn_bwd_dx with HFTT_SL_X3_F16 or HFTT_SL_X3_BF16).
//
// Geometry as in strip_gemm2.hip: one persistent workgroup per CU, 4 waves (one per SIMD, up to 512 registers), a wave owns a strip of 32
// tokens and keeps it in registers as the MFMA B operand for the whole block; the weights are the A operand, pre-packed in consumption
// order (hftt_x3_strip_pack) and streamed L2 -> LDS by LDS-DMA through a 4 x 16 KB ring, one s_barrier per slot.  What is different:
//   * a strip chunk (8 features of the lane's token) lives in 8 registers either as 8 raw fp32 values or, after conversion, as its hi
//     and lo halves (4 + 4 registers): conversion happens in place one slot before the chunk's first use, and as soon as a chunk has fed
//     its last MFMA of the block its registers receive the next block's values (two 16-byte loads) -- a rolling prefetch with no second
//     register set, so strip (128) + accumulators (128) leave room for everything else;
//   * a ring slot holds the hi and lo fragments of 8 output tiles for ONE k chunk: 16 fragments = 24 MFMAs per wave (lo.hi, hi.lo, hi.hi
//     per tile), i.e. 1.5 MFMAs per kilobyte read from LDS and per kilobyte streamed from L2 against 1.0 in the bf16 kernels;
//   * results leave from the epilogue directly (fp32, 16 bytes per lane).
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "hftt_common.h"
#include "x3_common.h"
#include "hftt_host.h"
#include "x3_internal.h"
#include "../../include/hftt_hip.h"

// Timing switches (HFTT_X3_DEBUG, results garbage) exist in the ablation build only (-DHFTT_X3_STRIP_ABLATE, tools/bench_x3.py): as run-time
// tests inside the slot loops they are not free (the same switches cost the attention backward 24 %).
#ifdef HFTT_X3_STRIP_ABLATE
#define XDBG(g, bit) (((g).pad & (bit)) != 0)
#else
#define XDBG(g, bit) false
#endif

namespace {

#include "strip_pipe.h"

__host__ __device__ inline int x3_i_of_c(int c) { const int g = c & 15, hh = c >> 4; return (g & 3) + 8 * (g >> 2) + 4 * hh; }

// ---------------------------------------------------------------------------------------------------------------------
// weight packing: one thread per 16-byte destination chunk pair (8 consecutive k of one logical row -> hi fragment piece + lo fragment piece)
// Fragment (tile, chunk, plane), 64 lanes x 8 halves: lane (i = lane & 31, hk = lane >> 5) holds, for chunk = 2*pt + u,
//   Wl[32*tile + c(i)][32*pt + 16*hk + 8*u + 0..7]   (c(i) as in the bf16 strip pack: accumulator register g of a lane = feature 16h + g).
// Slots of 16 fragments.  order 0 ("linear"): slot = (tile >> 3) * (K / 16) + chunk, fragment = 2 * (tile & 7) + plane;
// order 1 ("tile-major", K == 256): slot = 2 * tile + (chunk >> 3), fragment = 2 * (chunk & 7) + plane.
// Stream position of a slot: slot_offset + slot_stride * (slot >> 1) + (slot & 1)   (slots come in pairs; the fused FFN interleaves pairs).
// order 2 ("compact", small widths: x3s_strip.h): no slots -- the (hi, lo) pair of (chunk, tile) at pair index slot_offset + chunk * NT + tile.
// ---------------------------------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void x3_strip_pack_kernel(const float* __restrict__ params, unsigned short* __restrict__ dst,
                                                            const hftt_strip_pack_entry* __restrict__ table) {
  const hftt_strip_pack_entry e = table[blockIdx.y];
  const int nrows = e.transpose ? e.cols : e.rows;
  const int nk = e.transpose ? e.rows : e.cols;
  const int kch = nk >> 3;
  const long total = (long)nrows * kch;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int nl = (int)(t / kch), kc = (int)(t - (long)nl * kch);
    const int n = e.n0 + nl, k = e.k0 + kc * 8;
    float v[8];
    const float* src = params + e.src_off;
    if (e.transpose) {
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = src[(long)(kc * 8 + q) * e.src_ld + nl];
    } else {
      const float4 a = *reinterpret_cast<const float4*>(src + (long)nl * e.src_ld + kc * 8);
      const float4 b = *reinterpret_cast<const float4*>(src + (long)nl * e.src_ld + kc * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    const int tile = n >> 5, i = x3_i_of_c(n & 31);
    // feature k of the logical row sits at chunk 2*pt + u with pt = k >> 5, lane half hk = (k >> 4) & 1, u = (k >> 3) & 1
    const int pt = k >> 5, hk2 = (k >> 4) & 1, u = (k >> 3) & 1;
    const int ch = 2 * pt + u;
    long off;
    if (e.order == 2) {                                 // compact (x3s_strip.h): pair (chunk, tile) at slot_offset + chunk * NT + tile, NT = slot_stride
      off = (e.slot_offset + (long)ch * e.slot_stride + tile) * 1024 + (hk2 * 32 + i) * 8;
    } else {
      long slot, frag;
      if (e.order == 0) { slot = (long)(tile >> 3) * (e.K >> 4) + ch; frag = 2 * (tile & 7); }
      else { slot = 2 * tile + (ch >> 3); frag = 2 * (ch & 7); }
      const long pos = e.slot_offset + (long)e.slot_stride * (slot >> 1) + (slot & 1);
      off = (pos * 16 + frag) * 512 + (hk2 * 32 + i) * 8;
    }
    bf16x8 hi, lo;
    x3_split8_checked<E>(v, hi, lo);                  // (a NaN / Inf parameter poisons its fragment pair)
    *reinterpret_cast<bf16x8*>(dst + e.dst_off + off) = hi;
    *reinterpret_cast<bf16x8*>(dst + e.dst_off + off + 512) = lo;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------------
// the weight ring: stream fetched three slots ahead and ALWAYS (past the workgroup's last slot it wraps to the head: three harmless
// extra slots, drained before the kernel ends).  The barrier that opens slot s vouches for slot s + 1 (not s): every wave has waited for its
// quarter of fill(s + 1) -- "at least four vector-memory instructions were issued after it" (fill(s + 2)) holds at every slot boundary, so
// the wait is the constant s_waitcnt vmcnt(4) -- which lets the last two steps of slot s already request the first four fragments of slot
// s + 1 (x3_slot_*: `pre`).  (With the barrier vouching for slot s itself those four reads came right behind it: their LDS round trip, a few
// hundred cycles with nothing to issue, was paid four times per hidden tile by a wave alone on its SIMD.)  The refill issued in slot s
// overwrites the buffer of slot s - 1, which every wave has left before it passed the barrier of slot s.
struct XPipe {
  const unsigned short* w;
  int S;                        // slots per block
  int fill_pos;
  unsigned ring;
  int wave, lane;
  bool nofill, nobar;           // timing experiments only (HFTT_X3_DEBUG bits 0 / 1): results are garbage
  __device__ __forceinline__ const unsigned short* src_of(int pos) const { return w + ((long)pos * 16 + wave * 4) * 512 + lane * 8; }
  __device__ __forceinline__ void advance() { fill_pos = (fill_pos + 1 == S) ? 0 : fill_pos + 1; }
  template <int BUF>
  __device__ __forceinline__ void fill() {
    glds16x4(src_of(fill_pos), ring + (unsigned)BUF * SLOT_BYTES + (unsigned)wave * 4096u);
    advance();
  }
  __device__ __forceinline__ void begin_slot() {
    if (!nobar) { HFTT_WAITVM(4); __builtin_amdgcn_s_barrier(); }
  }
  // before the first slot of the kernel: slots 0 AND 1 complete in every wave (fill(2) may still be in flight)
  __device__ __forceinline__ void prologue_sync() { HFTT_WAITVM(4); __builtin_amdgcn_s_barrier(); }
  template <int BUF>
  __device__ __forceinline__ void refill() {      // BUF: the slot being consumed; the refill goes to (BUF + 3) % 4
    if (!nofill) glds16x4(src_of(fill_pos), ring + (unsigned)((BUF + FILL_AHEAD) & (NSLOT - 1)) * SLOT_BYTES + (unsigned)wave * 4096u);
    advance();
  }
  __device__ __forceinline__ void drain() { HFTT_WAITVM(0); }
};

// a strip chunk: 8 features of the lane's token in 8 registers -- raw fp32 (a = first four, b = last four) or converted (a = hi, b = lo)
struct XChunk { u4v a, b; };
// Streaming accesses of the strip kernels and the non-temporal hint (-DHFTT_STRIP_NT=0 none, 1 loads only, 2 result stores only = the default:
// round 6, same box, three interleaved runs each way, 280.8 against 278.3 clips/s -- profiles/r06_ab_nontemporal_result_stores.txt).  Why: the
// request-size counters (profiles/r04a_pmc_fetch_detail.json: all 128-byte requests, so FETCH_SIZE x 2 is exact here) show the QKV projection
// fetching 1.58 x its activation bytes and the training FFN 1.49 x -- lines evicted from the 4 MB L2 between their uses (the weight stream
// every 128-token block re-reads; 16-byte pieces of an activation line taken by different instructions) and fetched again from the fabric,
// i.e. from the 256 MB Infinity Cache.  Measured with the hint on every strip load and result store (same box, paper-size step): the FFN's
// fetch fell (225 -> 196 MB) but the QKV projection's rose (250 -> 283 MB) and the K = 768 dX form's rose from 698 to 909 MB -- a
// non-temporal load does not keep the line for the next instruction's pieces -- and the step went from 266 to 251 clips/s.  A result store
// has no such second use inside the kernel (its consumer is a later launch, tens of MB of other traffic away), so the hint only spares L2 lines.
#ifndef HFTT_STRIP_NT
#define HFTT_STRIP_NT 2
#endif
__device__ __forceinline__ u4v stream_load16(const void* p) {
#if HFTT_STRIP_NT == 1                               // (2: the hint on the result stores only)
  return __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
#else
  return *reinterpret_cast<const u4v*>(p);
#endif
}
__device__ __forceinline__ void stream_store16(void* p, const u4v& v) {
#if HFTT_STRIP_NT
  __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(p));
#else
  *reinterpret_cast<u4v*>(p) = v;
#endif
}
__device__ __forceinline__ void chunk_load(XChunk& c, const float* p) {
  c.a = stream_load16(p);
  c.b = stream_load16(p + 4);
}
template <int E>
__device__ __forceinline__ void chunk_convert(XChunk& c) {
  const float v[8] = {__uint_as_float(c.a.x), __uint_as_float(c.a.y), __uint_as_float(c.a.z), __uint_as_float(c.a.w),
                      __uint_as_float(c.b.x), __uint_as_float(c.b.y), __uint_as_float(c.b.z), __uint_as_float(c.b.w)};
  bf16x8 hi, lo;
  if (E == X3_BF16H) {                                 // hi half only (c.b is not read in this form)
    uint4 h;
    h.x = X3<E>::pk(v[0], v[1]); h.y = X3<E>::pk(v[2], v[3]); h.z = X3<E>::pk(v[4], v[5]); h.w = X3<E>::pk(v[6], v[7]);
    c.a = __builtin_bit_cast(u4v, h);
    return;
  }
  x3_split8<E>(v, hi, lo);
  c.a = __builtin_bit_cast(u4v, hi); c.b = __builtin_bit_cast(u4v, lo);
}
// a RAW chunk (elements 8 q .. 8 q + 7 of a dropout site's tensor) times its keep mask: the backward of a dropout whose output this strip is the
// gradient of, applied as the strip arrives (the LayerNorm backward then writes no masked copy of its result)
__device__ __forceinline__ void chunk_drop(XChunk& c, uint64_t seed, uint32_t site, uint64_t q0, uint32_t thr, float inv_keep) {
  float v[8] = {__uint_as_float(c.a.x), __uint_as_float(c.a.y), __uint_as_float(c.a.z), __uint_as_float(c.a.w),
                __uint_as_float(c.b.x), __uint_as_float(c.b.y), __uint_as_float(c.b.z), __uint_as_float(c.b.w)};
  drop8(v, seed, site, q0, thr, inv_keep);
  c.a = u4v{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  c.b = u4v{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
}
// hi + lo of a converted chunk back to fp32 (the residual of the fused FFN: the block input itself, to 2^-22)
template <int E>
__device__ __forceinline__ void chunk_values(const XChunk& c, float* v) {
  const unsigned h[4] = {c.a.x, c.a.y, c.a.z, c.a.w}, l[4] = {c.b.x, c.b.y, c.b.z, c.b.w};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    float h0, h1, l0, l1;
    X3<E>::unpk(h[q], h0, h1); X3<E>::unpk(l[q], l0, l1);
    v[2 * q] = h0 + l0; v[2 * q + 1] = h1 + l1;
  }
}
// element offset, inside the lane's row view (row + 16 * h), of chunk ch = 2 * pt + u: features 32 * pt + 8 * u + [0, 8) of the lane half
__device__ __forceinline__ int chunk_off(int ch) { return (ch >> 1) * 32 + (ch & 1) * 8; }

// One ring slot against ONE strip chunk: fragments 2t (hi) and 2t + 1 (lo) of output tile t -> acc[t] += lo.x_hi + hi.x_lo + hi.x_hi.
// Four fragment reads are in flight ahead of the MFMAs; side(i) (i = 0 .. 23) is called behind MFMA i, so the slot's memory and
// conversion work sits in program order BETWEEN the MFMAs (a wave alone on its SIMD issues in order).
// pre: fragments 0 .. 3 of THIS slot on entry (requested by the previous slot, or by slot_head() at the head of a block), of the NEXT slot
// (`next`, already vouched for by this slot's barrier: XPipe) on exit.
struct XPre { bf16x8 f[4]; bool noread = false; };     // noread: timing experiments only (HFTT_X3_DEBUG bit 2048: fragments are not re-read)
__device__ __forceinline__ void slot_head(const unsigned char* slot, XPre& pre) {
#pragma unroll
  for (int i = 0; i < 4; i++) pre.f[i] = *reinterpret_cast<const bf16x8*>(slot + i * 1024);
}
template <int E, typename G>
__device__ __forceinline__ void x3_slot_tiles(const unsigned char* slot, const unsigned char* next, XPre& pre, const XChunk& x, f32x16 (&acc)[8], G&& side) {
  bf16x8 fr[16];
#pragma unroll
  for (int i = 0; i < 4; i++) fr[i] = pre.f[i];
  const bf16x8 xh = __builtin_bit_cast(bf16x8, x.a), xl = __builtin_bit_cast(bf16x8, x.b);
  static_for<8>([&](auto t_c) __attribute__((always_inline)) {
    constexpr int t = decltype(t_c)::value;
    acc[t] = X3<E>::mma(fr[2 * t + 1], xh, acc[t]);
    side(std::integral_constant<int, 3 * t>{});
    if (pre.noread) { if (2 * t + 4 < 16) fr[2 * t + 4] = fr[2 * t]; }
    else if (2 * t + 4 < 16) fr[2 * t + 4] = *reinterpret_cast<const bf16x8*>(slot + (2 * t + 4) * 1024);
    else pre.f[2 * t + 4 - 16] = *reinterpret_cast<const bf16x8*>(next + (2 * t + 4 - 16) * 1024);
    if (E != X3_BF16H) acc[t] = X3<E>::mma(fr[2 * t], xl, acc[t]);      // (X3_BF16H: the strip is a gradient, hi half only)
    side(std::integral_constant<int, 3 * t + 1>{});
    if (pre.noread) { if (2 * t + 5 < 16) fr[2 * t + 5] = fr[2 * t + 1]; }
    else if (2 * t + 5 < 16) fr[2 * t + 5] = *reinterpret_cast<const bf16x8*>(slot + (2 * t + 5) * 1024);
    else pre.f[2 * t + 5 - 16] = *reinterpret_cast<const bf16x8*>(next + (2 * t + 5 - 16) * 1024);
    acc[t] = X3<E>::mma(fr[2 * t], xh, acc[t]);
    side(std::integral_constant<int, 3 * t + 2>{});
    __builtin_amdgcn_sched_barrier(0);
  });
}
// One ring slot against EIGHT strip chunks (tile-major fc_1 half tile): fragments 2q (hi), 2q + 1 (lo) of chunk x[q] -> one accumulator
template <int E, int OFF, typename G>
__device__ __forceinline__ void x3_slot_chunks(const unsigned char* slot, const unsigned char* next, XPre& pre, const XChunk (&x)[16], f32x16& acc, G&& side) {
  bf16x8 fr[16];
#pragma unroll
  for (int i = 0; i < 4; i++) fr[i] = pre.f[i];
  static_for<8>([&](auto q_c) __attribute__((always_inline)) {
    constexpr int q = decltype(q_c)::value;
    const bf16x8 xh = __builtin_bit_cast(bf16x8, x[OFF + q].a), xl = __builtin_bit_cast(bf16x8, x[OFF + q].b);
    acc = X3<E>::mma(fr[2 * q + 1], xh, acc);
    side(std::integral_constant<int, 3 * q>{});
    if (pre.noread) { if (2 * q + 4 < 16) fr[2 * q + 4] = fr[2 * q]; }
    else if (2 * q + 4 < 16) fr[2 * q + 4] = *reinterpret_cast<const bf16x8*>(slot + (2 * q + 4) * 1024);
    else pre.f[2 * q + 4 - 16] = *reinterpret_cast<const bf16x8*>(next + (2 * q + 4 - 16) * 1024);
    if (E != X3_BF16H) acc = X3<E>::mma(fr[2 * q], xl, acc);
    side(std::integral_constant<int, 3 * q + 1>{});
    if (pre.noread) { if (2 * q + 5 < 16) fr[2 * q + 5] = fr[2 * q + 1]; }
    else if (2 * q + 5 < 16) fr[2 * q + 5] = *reinterpret_cast<const bf16x8*>(slot + (2 * q + 5) * 1024);
    else pre.f[2 * q + 5 - 16] = *reinterpret_cast<const bf16x8*>(next + (2 * q + 5 - 16) * 1024);
    acc = X3<E>::mma(fr[2 * q], xh, acc);
    side(std::integral_constant<int, 3 * q + 2>{});
    __builtin_amdgcn_sched_barrier(0);
  });
}

__device__ __forceinline__ void load16f(const float* p, float* v) {
#pragma unroll
  for (int q = 0; q < 4; q++) { const float4 t = reinterpret_cast<const float4*>(p)[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}
__device__ __forceinline__ void store16f(float* p, const float* v) {
#pragma unroll
  for (int q = 0; q < 4; q++) reinterpret_cast<float4*>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// A wave's 32 x 32 fp32 tile (lane (j, h): token j, columns 16h .. 16h+15 in v) -> global memory as WHOLE 128-byte row segments: 16 rows at a
// time through a wave-private LDS patch (16 rows x 36 floats), read back with 8 lanes per row, so one store instruction writes 8 complete
// lines.  Stored straight from the accumulator layout an instruction scatters 64 16-byte pieces over 32 rows, and a CU's store path then
// moves ~7 bytes per clock (MI355X_MICROARCH.md, store-issue bound): the fp32 results of the QKV projection cost 210 us of its 510 us that
// way (HFTT_X3_DEBUG=16).  LDS operations of one wave execute in order, so the patch needs no barrier.
constexpr int STG_RS = 36;                            // floats per staged row (144 B: the 16-byte row pieces of 8 rows fall on 8 different bank quads)
constexpr int STG_BYTES_PER_WAVE = 16 * STG_RS * 4;
// Addresses: gtile and the row-group offset (16 * half + 8 * k) * ld are wave-uniform (scalar registers, scalar multiplies); the lane's own part
// is the 32-bit element offset ro = (lane >> 3) * ld + 4 * (lane & 7), one full-rate 24-bit multiply per call.  (Formed as 64-bit (row * ld) per
// lane and store, the epilogue of the fc_o + LayerNorm kernel spent 236 v_mul_lo_u32 and 114 v_mad_u64_u32 -- quarter-rate -- on addresses.)
__device__ __forceinline__ void tile_store_rows(float* stage, const float* v, int j, int h, int lane, float* gtile, long ld, bool ok) {
  asm volatile("" : "+v"(lane));                    // (formed per call: hoisted out of the block loop the per-lane offsets of every tensor are spilled)
  const unsigned ro = __umul24((unsigned)lane >> 3, (unsigned)ld) + ((unsigned)lane & 7u) * 4u;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if ((j >> 4) == half) {
      float* w = stage + (j & 15) * STG_RS + 16 * h;
#pragma unroll
      for (int q = 0; q < 4; q++) reinterpret_cast<float4*>(w)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int r = (lane >> 3) + 8 * k;
      const float4 t = *reinterpret_cast<const float4*>(stage + r * STG_RS + (lane & 7) * 4);
      if (ok) stream_store16(gtile + (long)(half * 16 + 8 * k) * ld + ro, __builtin_bit_cast(u4v, t));
    }
  }
}

// The same tile as bf16 (the saved FFN hidden and its gradient: operands of the weight-gradient products only, HFTT_SL_H_BF16): 64-byte row
// segments, 16 rows x 4 lanes per store instruction, through the same patch (rows of 40 shorts = 80 B: 16-byte pieces, 8 rows on 8 bank quads).
__device__ __forceinline__ void tile_store_rows_bf16(float* stage, const float* v, int j, int h, int lane, unsigned short* gtile, long ld, bool ok) {
  asm volatile("" : "+v"(lane));                    // (formed per call, as in tile_store_rows)
  const unsigned ro = __umul24((unsigned)lane >> 2, (unsigned)ld) + ((unsigned)lane & 3u) * 8u;
  unsigned short* st16 = reinterpret_cast<unsigned short*>(stage);
  constexpr int RS = 40;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if ((j >> 4) == half) {
      unsigned short* w = st16 + (j & 15) * RS + 16 * h;
      uint4 a, b;
      a.x = f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); a.y = f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
      a.z = f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16); a.w = f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
      b.x = f2bf(v[8]) | ((unsigned)f2bf(v[9]) << 16); b.y = f2bf(v[10]) | ((unsigned)f2bf(v[11]) << 16);
      b.z = f2bf(v[12]) | ((unsigned)f2bf(v[13]) << 16); b.w = f2bf(v[14]) | ((unsigned)f2bf(v[15]) << 16);
      reinterpret_cast<uint4*>(w)[0] = a;
      reinterpret_cast<uint4*>(w)[1] = b;
    }
    const int r = lane >> 2;
    const uint4 t = *reinterpret_cast<const uint4*>(st16 + r * RS + (lane & 3) * 8);
    if (ok) stream_store16(gtile + (long)(half * 16) * ld + ro, __builtin_bit_cast(u4v, t));
  }
}
// The same tile as an f16 PAIR (HFTT_SL_C_F16PAIR: the q / k / v projections, read by the attention kernels as MFMA operands without any
// further conversion -- x3_attn_pl.hip): the tile's 128 bytes per row hold its 32 hi halves (64 B), then its 32 lo halves -- the same
// whole-line row segments as the fp32 form, through the same patch.
__device__ __forceinline__ void tile_store_rows_f16pair(float* stage, const float* v, int j, int h, int lane, float* gtile, long ld, bool ok) {
  asm volatile("" : "+v"(lane));                    // (formed per call, as in tile_store_rows)
  const unsigned ro = __umul24((unsigned)lane >> 3, (unsigned)ld) + ((unsigned)lane & 7u) * 4u;
  bf16x8 hi0, lo0, hi1, lo1;
  x3_split8<X3_F16>(v, hi0, lo0);
  x3_split8<X3_F16>(v + 8, hi1, lo1);
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if ((j >> 4) == half) {
      unsigned char* w = reinterpret_cast<unsigned char*>(stage + (j & 15) * STG_RS) + 32 * h;
      *reinterpret_cast<bf16x8*>(w) = hi0;
      *reinterpret_cast<bf16x8*>(w + 16) = hi1;
      *reinterpret_cast<bf16x8*>(w + 64) = lo0;
      *reinterpret_cast<bf16x8*>(w + 80) = lo1;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int r = (lane >> 3) + 8 * k;
      const float4 t = *reinterpret_cast<const float4*>(stage + r * STG_RS + (lane & 7) * 4);
      if (ok) stream_store16(gtile + (long)(half * 16 + 8 * k) * ld + ro, __builtin_bit_cast(u4v, t));
    }
  }
}
__device__ __forceinline__ void load16h(const unsigned short* p, float* v) {      // 16 bf16 -> fp32
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const uint4 t = reinterpret_cast<const uint4*>(p)[q];
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; e++) { v[8 * q + 2 * e] = bf2f((unsigned short)(w[e] & 0xFFFFu)); v[8 * q + 2 * e + 1] = bf2f((unsigned short)(w[e] >> 16)); }
  }
}

// LayerNorm over the 256 features of the lane's token (lane: 128 of them in acc, partner lane ^ 32 the rest), fp32 out
// KEEP: the normalised rows stay in acc (the fused fc_o + FFN kernel feeds them to the next GEMM); y_wave may then be NULL (nothing stored)
template <bool KEEP = false>
__device__ __forceinline__ void x3_ln_rows(f32x16 (&acc)[8], const float* gamma_lds, const float* beta_lds, int h, float* mean_out, float* rstd_out,
                                           long tok, bool ok, float* stage, int j, int lane, float* pre_wave, float* y_wave, long ld,
                                           unsigned short* pre16_wave = nullptr) {
  float s = 0.f;
#pragma unroll
  for (int ot = 0; ot < 8; ot++)
#pragma unroll
    for (int q = 0; q < 16; q++) s += acc[ot][q];
  const float mean = xor32_sum(s) * (1.0f / 256.0f);
  float qs = 0.f;
#pragma unroll
  for (int ot = 0; ot < 8; ot++)
#pragma unroll
    for (int q = 0; q < 16; q++) { const float dlt = acc[ot][q] - mean; qs += dlt * dlt; }
  const float rstd = 1.0f / sqrtf(xor32_sum(qs) * (1.0f / 256.0f) + 1e-5f);
  if (ok && h == 0) {
    if (mean_out != nullptr) mean_out[tok] = mean;
    if (rstd_out != nullptr) rstd_out[tok] = rstd;
  }
#pragma unroll
  for (int ot = 0; ot < 8; ot++) {
    float v[16], ga[16], be[16];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = acc[ot][q];
    if (pre_wave != nullptr) tile_store_rows(stage, v, j, h, lane, pre_wave + ot * 32, ld, ok);      // (wave-uniform pointer tests)
    if (pre16_wave != nullptr) tile_store_rows_bf16(stage, v, j, h, lane, pre16_wave + ot * 32, ld, ok);   // HFTT_SL_PRE_BF16: read by the LayerNorm backward only
    lds16f(gamma_lds + ot * 32 + 16 * h, ga);
    lds16f(beta_lds + ot * 32 + 16 * h, be);
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = (v[q] - mean) * rstd * ga[q] + be[q];
    if (!KEEP || y_wave != nullptr) tile_store_rows(stage, v, j, h, lane, y_wave + ot * 32, ld, ok);
    if (KEEP) {
#pragma unroll
      for (int q = 0; q < 16; q++) acc[ot][q] = v[q];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// C = epi(x . Wl^T + bias), fp32 tensors, K = 256 * KCH, N = 256 * PASSES; LN: N == 256 with dropout / residual / LayerNorm
// ---------------------------------------------------------------------------------------------------------------------
// XR: strip chunks resident per wave.  16 = a whole 256-feature set (one wave per SIMD, 512 registers); 8 = half a set (PASSES == 1, no LayerNorm):
// chunk c + 8 is loaded into the registers of chunk c one slot after it was consumed, the kernel fits 256 registers and TWO workgroups share a CU.
template <int E, bool LN, int PASSES, int KCH, bool HR, int XR = 16>
__global__ __launch_bounds__(256, XR == 16 ? 1 : 2) void x3_linear_kernel(const hftt_strip_desc g) {
  static_assert(XR == 16 || (XR == 8 && PASSES == 1 && !LN), "the half-set form reloads every set (one pass); the LayerNorm epilogue does not fit 256 registers");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  constexpr int STEPS = PASSES * KCH;               // 16-slot steps per block
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // bias[N] | gamma[256] | beta[256] | four wave-private store patches
  float* stage = reinterpret_cast<float*>(smem + RING_BYTES + 4 * (PASSES * 256 + 512) + wave * STG_BYTES_PER_WAVE);
  const float* xb = reinterpret_cast<const float*>(g.x);
  float* cb = reinterpret_cast<float*>(g.C);
  float* preb = reinterpret_cast<float*>(g.pre_ln_out);
  const float* rb = reinterpret_cast<const float*>(g.residual);
  const bool relu = g.flags & HFTT_SL_RELU;

  XPipe P;
  P.w = g.w; P.S = STEPS * 16; P.fill_pos = 0;
  P.ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  P.wave = wave; P.lane = lane;
  P.nofill = XDBG(g, 1); P.nobar = XDBG(g, 2);
  const bool dbg_nostore = XDBG(g, 16), dbg_noload = XDBG(g, 32), dbg_noconv = XDBG(g, 64);

  auto tok_of = [&](long blk) { const long t = blk * 128 + wave * 32 + j; return t < g.M ? t : (long)g.M - 1; };

  for (int i = tid; i < g.N; i += 256) prm[i] = g.bias != nullptr ? g.bias[i] : 0.f;
  if (LN) { prm[g.N + tid] = g.ln_gamma[tid]; prm[g.N + 256 + tid] = g.ln_beta[tid]; }
  XChunk xr[XR];
  {
    const float* p0 = xb + tok_of(blockIdx.x) * g.ldx + 16 * h;
#pragma unroll
    for (int c = 0; c < XR; c++) chunk_load(xr[c], p0 + chunk_off(c));
  }
  P.fill<0>(); P.fill<1>(); P.fill<2>();
  static_assert(FILL_AHEAD == 3, "prologue fills");
  wait_lgkm0();
  P.prologue_sync();                       // the parameter rows in LDS are read (by every wave) before the first slot's barrier
  chunk_convert<E>(xr[0]);

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;

  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;                                       // opaque per iteration (LICM would hoist per-tile column arithmetic out of the loop: spills)
    asm volatile("" : "+v"(hb));
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;          // M % 32 == 0 (host check)
    const long nxt = blk + gridDim.x;
    // per-lane row numbers are formed where a step uses them, from a lane number the optimiser cannot see through: as loop-level values
    // (this block's and the next block's strip rows, the residual row, the dropout row index, 64 bits each) they were spilled to scratch
    auto lane_tok = [&](long b) { int l_ = lane; asm volatile("" : "+v"(l_)); const long t = b * 128 + wave * 32 + (l_ & 31); return t < g.M ? t : (long)g.M - 1; };
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const float* prm_b = prm + zero;                  // (keeps the LDS parameter reads inside the iteration)
    for (int pass = 0; pass < PASSES; pass++) {
      f32x16 acc[8];
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {                // accumulators start from the bias
        float b[16];
        lds16f(prm_b + pass * 256 + ot * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) acc[ot][q] = b[q];
      }
      XPre pre;
      slot_head(abase, pre);                          // (every pass starts at ring buffer 0; inside the pass the slots hand `pre` on)
      for (int kc = 0; kc < KCH; kc++) {
        // the chunk set in the registers at this step is k chunk kc; as a chunk is consumed its registers receive the same chunk position
        // of the NEXT step's set (next k chunk, or k chunk 0 of the next pass / block).  KCH == 1: the set is re-used by every pass of
        // the block and only replaced (by the next block's) during the last pass; conversion only in pass 0.
        const bool last_step = (pass == PASSES - 1) && (kc == KCH - 1);
        const bool reload = ((KCH > 1) || last_step) && !dbg_noload;
        const bool convert = ((KCH > 1) || (pass == 0)) && !dbg_noconv;
        // (past the last block: a harmless re-read of this block's rows)
        const float* nsrc = xb + lane_tok(last_step ? (nxt < nblk ? nxt : blk) : blk) * g.ldx + 16 * hb + (last_step ? 0 : ((kc + 1 == KCH) ? 0 : kc + 1) * 256);
        const float* csrc = (XR == 16) ? nsrc : xb + lane_tok(blk) * g.ldx + 16 * hb + kc * 256;       // XR == 8: the second half of THIS step's set
        static_for<16>([&](auto c_c) __attribute__((always_inline)) {
          constexpr int c = decltype(c_c)::value;
          constexpr int BUF = c & 3;
          P.begin_slot();
          x3_slot_tiles<E>(abase + BUF * SLOT_BYTES, abase + ((BUF + 1) & 3) * SLOT_BYTES, pre, xr[c % XR], acc, [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            if (i == 1) P.template refill<BUF>();
            if (XR == 16) {
              if (i == 5 && c > 0) { if (reload) chunk_load(xr[(c - 1) % XR], nsrc + chunk_off(c - 1)); }
            } else {                                  // the registers of chunk c - 1 take chunk c - 1 + XR (of this set, then of the next)
              if (i == 5 && c > 0) chunk_load(xr[(c - 1) % XR], (c - 1 + XR < 16) ? csrc + chunk_off((c - 1 + XR) & 15) : nsrc + chunk_off((c - 1 + XR) & 15));
            }
            if (i == 9 && c < 15) { if (convert) chunk_convert<E>(xr[(c + 1) % XR]); }
          });
        });
        // the last chunk of the set is replaced here, and chunk 0 of the next set converted (its load is 15 slots old; XR == 8: 7 slots)
        if (XR == 16) { if (reload) chunk_load(xr[15], nsrc + chunk_off(15)); }
        else chunk_load(xr[XR - 1], nsrc + chunk_off(XR - 1));
        if (((KCH > 1) || last_step) && !dbg_noconv) chunk_convert<E>(xr[0]);
      }
      // ---------------- epilogue of this pass ----------------
      const long tokc_e = lane_tok(blk);
      const long tok = blk * 128 + wave * 32 + (tokc_e & 31);                 // (M % 32 == 0: the clamp never changes the low five bits)
      const uint64_t rowq = ((uint64_t)tok * (uint64_t)g.N) >> 2;
      float* cwave = cb + (blk * 128 + wave * 32) * g.ldc + pass * 256;      // row 0 of this wave's strip, this pass's columns
      const long rrow = g.res_mod > 0 ? (long)((unsigned)tokc_e % (unsigned)g.res_mod) : tokc_e;
      const float* rrow_p = rb + (HR ? rrow * g.ldr + pass * 256 + 16 * hb : 0);
      float rnext[16];
      if (HR) load16f(rrow_p, rnext);
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
        const int col0 = pass * 256 + ot * 32 + 16 * hb;
        float v[16], r[16];
        if (HR) {
#pragma unroll
          for (int q = 0; q < 16; q++) r[q] = rnext[q];
          if (ot < 7) load16f(rrow_p + (ot + 1) * 32, rnext);
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
          float t = acc[ot][q];
          if (!LN && relu) t = fmaxf(t, 0.f);
          v[q] = t * g.out_scale;
        }
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.drop_site, rowq + (col0 >> 2), thr, inv_keep);
        if (HR) {
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] += r[q];
        }
        if (LN) {
#pragma unroll
          for (int q = 0; q < 16; q++) acc[ot][q] = v[q];
        } else {
          tile_store_rows(stage, v, j, hb, lane, cwave + ot * 32, g.ldc, wave_ok && !dbg_nostore);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LN) {
        const bool p16 = g.flags & HFTT_SL_PRE_BF16;
        x3_ln_rows(acc, prm_b + g.N, prm_b + g.N + 256, hb, g.ln_mean, g.ln_rstd, tok, wave_ok, stage, j, lane,
                   (preb != nullptr && !p16) ? preb + (blk * 128 + wave * 32) * g.ldc : nullptr, cwave, g.ldc,
                   (preb != nullptr && p16) ? reinterpret_cast<unsigned short*>(preb) + (blk * 128 + wave * 32) * g.ldc : nullptr);
      }
    }
  }
  P.drain();
}

// ---------------------------------------------------------------------------------------------------------------------
// C = epi(x . Wl^T + bias) for K == 256 without LayerNorm, output-tile-major ("N-outer"): one 32-column output tile at a time against the
// whole resident strip (two ring slots of a tile-major pack), so a wave carries ONE accumulator tile (16 registers) instead of eight, a
// tile's 64 bytes per lane leave as soon as its 48 MFMAs are done -- stores spread evenly over the block instead of one burst per pass --
// and the kernel fits 256 registers: TWO workgroups per CU (each with its own ring) run free of each other, one's epilogue, conversions and
// strip loads under the other's MFMAs.  (The K-outer form above, one wave per SIMD, ran MFMA, loads and stores strictly one after the
// other: QKV at 262,144 tokens 560 us = 250 us of MFMA + 240 us of stores + 120 us of loads, HFTT_X3_DEBUG.)
// ---------------------------------------------------------------------------------------------------------------------
// PLN: C leaves as f16-pair planes (HFTT_SL_C_F16PAIR; forward products only, no residual)
// XD: the strip is the gradient of a dropout output (HFTT_SL_X_DROP): masked while it is converted; no epilogue dropout
template <int E, int NT, bool HR, bool PLN = false, bool XD = false>
__global__ __launch_bounds__(256, 2) void x3_linear_n_kernel(const hftt_strip_desc g) {
  static_assert(!PLN || (E == X3_F16 && !HR), "f16-pair output: a forward projection without residual");
  static_assert(!XD || (E != X3_F16 && !PLN), "masked strip: a backward product");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // bias[N] | four wave-private store patches
  float* stage = reinterpret_cast<float*>(smem + RING_BYTES + 4 * NT * 32 + wave * STG_BYTES_PER_WAVE);
  const float* xb = reinterpret_cast<const float*>(g.x);
  float* cb = reinterpret_cast<float*>(g.C);
  const float* rb = reinterpret_cast<const float*>(g.residual);
  const bool relu = g.flags & HFTT_SL_RELU;

  XPipe P;
  P.w = g.w; P.S = 2 * NT; P.fill_pos = 0;
  P.ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  P.wave = wave; P.lane = lane;
  P.nofill = XDBG(g, 1); P.nobar = XDBG(g, 2);
  const bool dbg_nostore = XDBG(g, 16);

  auto tok_of = [&](long blk) { const long t = blk * 128 + wave * 32 + j; return t < g.M ? t : (long)g.M - 1; };
  for (int i = tid; i < g.N; i += 256) prm[i] = g.bias != nullptr ? g.bias[i] : 0.f;
  XChunk xr[16];
  {
    const float* p0 = xb + tok_of(blockIdx.x) * g.ldx + 16 * h;
#pragma unroll
    for (int c = 0; c < 16; c++) chunk_load(xr[c], p0 + chunk_off(c));
  }
  P.fill<0>(); P.fill<1>(); P.fill<2>();
  static_assert(FILL_AHEAD == 3, "prologue fills");
  wait_lgkm0();
  P.prologue_sync();

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;
    asm volatile("" : "+v"(hb));
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;
    const long nxt = blk + gridDim.x;
    // per-lane row pointers are formed where a tile uses them, from a lane number the optimiser cannot see through (kept live across the
    // tile loop -- the next block's strip rows, the residual rows, the dropout row index -- they were the kernel's spills: 2 .. 7 registers)
    auto lane_tok = [&](long b) { int l_ = lane; asm volatile("" : "+v"(l_)); const long t = b * 128 + wave * 32 + (l_ & 31); return t < g.M ? t : (long)g.M - 1; };
    float* cwave = cb + (blk * 128 + wave * 32) * g.ldc;        // row 0 of this wave's strip (wave-uniform)
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const float* prm_b = prm + zero;
    if (XD && g.drop_p > 0.f) {
      const uint64_t rowq_x = ((uint64_t)lane_tok(blk) * (uint64_t)g.K) >> 2;
#pragma unroll
      for (int c = 0; c < 16; c++) chunk_drop(xr[c], g.drop_seed, g.drop_site, rowq_x + ((chunk_off(c) + 16 * hb) >> 2), thr, inv_keep);
    }
#pragma unroll
    for (int c = 0; c < 16; c++) chunk_convert<E>(xr[c]);       // this block's strip (loaded during the previous block's last tile)
    XPre pre;
    slot_head(abase, pre);                                      // (every block starts at ring buffer 0)

    // a run-time loop over tile PAIRS (the ring buffers repeat every two tiles): fully unrolled, hipcc forms every tile's store / residual
    // address up front and spills them
    for (int tp = 0; tp < NT / 2; tp++)
    static_for<2>([&](auto hf_c) __attribute__((always_inline)) {
      constexpr int half = decltype(hf_c)::value;
      constexpr int BA = (2 * half) & 3, BB = (2 * half + 1) & 3;
      const int t = 2 * tp + half;
      const bool LAST = (t == NT - 1);
      f32x16 hacc;
      {
        float b[16];
        lds16f(prm_b + t * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) hacc[q] = b[q];
      }
      float r[16];
      P.begin_slot();
      x3_slot_chunks<E, 0>(abase + BA * SLOT_BYTES, abase + BB * SLOT_BYTES, pre, xr, hacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<BA>();
        if (i == 4 && HR) {
          const long tokc = lane_tok(blk);
          const long rrow = g.res_mod > 0 ? (long)((unsigned)tokc % (unsigned)g.res_mod) : tokc;
          load16f(rb + rrow * g.ldr + 16 * hb + t * 32, r);
        }
        // last tile of the block: a chunk that has fed its three MFMAs takes the next block's values
        if (LAST && i % 3 == 2) chunk_load(xr[i / 3], xb + lane_tok(nxt < nblk ? nxt : blk) * g.ldx + 16 * hb + chunk_off(i / 3));
      });
      P.begin_slot();
      x3_slot_chunks<E, 8>(abase + BB * SLOT_BYTES, abase + ((BB + 1) & 3) * SLOT_BYTES, pre, xr, hacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<BB>();
        if (LAST && i % 3 == 2) chunk_load(xr[8 + i / 3], xb + lane_tok(nxt < nblk ? nxt : blk) * g.ldx + 16 * hb + chunk_off(8 + i / 3));
      });
      // ---- this tile's epilogue ----
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        float a = hacc[q];
        if (relu) a = fmaxf(a, 0.f);
        v[q] = a * g.out_scale;
      }
      if (!XD && g.drop_p > 0.f) {
        int l_ = lane; asm volatile("" : "+v"(l_));
        const uint64_t rowq = ((uint64_t)(blk * 128 + wave * 32 + (l_ & 31)) * (uint64_t)g.N) >> 2;
        drop16(v, g.drop_seed, g.drop_site, rowq + ((t * 32 + 16 * hb) >> 2), thr, inv_keep);
      }
      if (HR) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] += r[q];
      }
      if (PLN) tile_store_rows_f16pair(stage, v, j, hb, lane, cwave + t * 32, g.ldc, wave_ok && !dbg_nostore);
      else tile_store_rows(stage, v, j, hb, lane, cwave + t * 32, g.ldc, wave_ok && !dbg_nostore);
    });
  }
  P.drain();
}

// ---------------------------------------------------------------------------------------------------------------------
// fused two-GEMM block, d = 256, p = 32 * PT: mode 0 = FFN forward + residual + LayerNorm (fp16 halves), mode 1 = dX half of its
// backward (bf16 halves).  Stream per hidden tile t: two slots of the first matrix (tile-major: k chunks 0-7, 8-15), then two slots of
// the second (K-slice t: u = 0, 1).
// ---------------------------------------------------------------------------------------------------------------------
// HH: h_out / gate are bf16 [M, p] (HFTT_SL_H_BF16) instead of fp32; GH (mode 1): the gradient strips (dy, dh) enter as their hi halves
template <int MODE, int PT, bool HH, bool GH>
__global__ __launch_bounds__(256, 1) void x3_mlp_kernel(const hftt_ffn_desc g) {
  constexpr int E = (MODE == 0) ? X3_F16 : (GH ? X3_BF16H : X3_BF16);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  constexpr int p = PT * 32;
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // b1[p] | b2[256] | gamma[256] | beta[256] | four wave-private store patches
  float* stage = reinterpret_cast<float*>(smem + RING_BYTES + 4 * (PT * 32 + 768) + wave * STG_BYTES_PER_WAVE);
  const float* xb = reinterpret_cast<const float*>(g.x);
  float* yb = reinterpret_cast<float*>(g.y);
  float* preb = reinterpret_cast<float*>(g.pre_ln_out);
  const float* rb = reinterpret_cast<const float*>(g.residual);
  float* hob = reinterpret_cast<float*>(g.h_out);
  const float* gtb = reinterpret_cast<const float*>(g.gate);
  unsigned short* hob16 = reinterpret_cast<unsigned short*>(g.h_out);
  const unsigned short* gtb16 = reinterpret_cast<const unsigned short*>(g.gate);
  const bool has_res = (MODE == 1) && g.residual != nullptr;

  XPipe P;
  P.w = g.w; P.S = 4 * PT; P.fill_pos = 0;
  P.ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  P.wave = wave; P.lane = lane;
  P.nofill = XDBG(g, 1); P.nobar = XDBG(g, 2);

  auto tok_of = [&](long blk) { const long t = blk * 128 + wave * 32 + j; return t < g.M ? t : (long)g.M - 1; };

  for (int i = tid; i < p; i += 256) prm[i] = (MODE == 0 && g.b1 != nullptr) ? g.b1[i] : 0.f;
  prm[p + tid] = (MODE == 0 && g.b2 != nullptr) ? g.b2[tid] : 0.f;
  if (MODE == 0) { prm[p + 256 + tid] = g.ln_gamma[tid]; prm[p + 512 + tid] = g.ln_beta[tid]; }

  XChunk xr[16];
  {
    const float* p0 = xb + tok_of(blockIdx.x) * g.ldx + 16 * h;
#pragma unroll
    for (int c = 0; c < 16; c++) chunk_load(xr[c], p0 + chunk_off(c));
  }
  P.fill<0>(); P.fill<1>(); P.fill<2>();
  static_assert(FILL_AHEAD == 3, "prologue fills");
  wait_lgkm0();
  P.prologue_sync();

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;
    asm volatile("" : "+v"(hb));
    const long tok = blk * 128 + wave * 32 + j;
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;
    const long tokc = tok_of(blk);
    const long nxt = blk + gridDim.x;
    const float* xrow_next = xb + (nxt < nblk ? tok_of(nxt) : tokc) * g.ldx + 16 * hb;
    const uint64_t rowq_h = ((uint64_t)tok * (uint64_t)p) >> 2;
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const float* prm_b = prm + zero;
    // this block's strip: loaded by the previous block's epilogue (or the prologue), converted here.  Mode 1 with drop_p > 0: dy is the gradient
    // of the block's OUTPUT dropout (site_o) and is masked here (the LayerNorm backward in front writes no masked copy)
    if (MODE == 1 && g.drop_p > 0.f) {
      const uint64_t rowq_x = ((uint64_t)tokc * 256ull) >> 2;
#pragma unroll
      for (int c = 0; c < 16; c++) chunk_drop(xr[c], g.drop_seed, g.site_o, rowq_x + ((chunk_off(c) + 16 * hb) >> 2), thr, inv_keep);
    }
#pragma unroll
    for (int c = 0; c < 16; c++) chunk_convert<E>(xr[c]);

    f32x16 yacc[8];
#pragma unroll
    for (int ot = 0; ot < 8; ot++) {
      float b[16];
      lds16f(prm_b + p + ot * 32 + 16 * hb, b);
#pragma unroll
      for (int q = 0; q < 16; q++) yacc[ot][q] = b[q];
    }
    float gnext[16];                                  // mode 1: stored hidden (the gate) of the next tile
    if (MODE == 1) { if (HH) load16h(gtb16 + tokc * g.ldg + 16 * hb, gnext); else load16f(gtb + tokc * g.ldg + 16 * hb, gnext); }

    XPre pre;
    pre.noread = XDBG(g, 2048);
    slot_head(abase, pre);
    // (a run-time loop: every hidden tile uses ring buffers 0..3 in order, and the body is ~100 MFMAs -- unrolled 16 times the kernel was
    // 14,000 instructions, far beyond the instruction cache)
    for (int t = 0; t < PT; t++) {
      // ---- first GEMM, hidden tile t: the whole strip against two slots ----
      f32x16 hacc;
      {
        float b[16];
        lds16f(prm_b + t * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) hacc[q] = b[q];
      }
      float gcur[16];
      if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 16; q++) gcur[q] = gnext[q];
      }
      P.begin_slot();
      if (!XDBG(g, 256))
      x3_slot_chunks<E, 0>(abase + 0 * SLOT_BYTES, abase + 1 * SLOT_BYTES, pre, xr, hacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<0>();
        if (i == 6 && MODE == 1 && t + 1 < PT) {
          if (HH) load16h(gtb16 + tokc * g.ldg + (t + 1) * 32 + 16 * hb, gnext); else load16f(gtb + tokc * g.ldg + (t + 1) * 32 + 16 * hb, gnext);
        }
      });
      P.begin_slot();
      if (!XDBG(g, 256))
      x3_slot_chunks<E, 8>(abase + 1 * SLOT_BYTES, abase + 2 * SLOT_BYTES, pre, xr, hacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<1>();
      });
      // ---- middle epilogue: the lane's 16 hidden features of tile t become the B operand of the second GEMM ----
      float v[16];
      const int hcol0 = t * 32 + 16 * hb;
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = fmaxf(hacc[q], 0.f);
        if (g.drop_p > 0.f && !XDBG(g, 128)) drop16(v, g.drop_seed, g.site_h, rowq_h + (hcol0 >> 2), thr, inv_keep);
      } else {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = gcur[q] > 0.f ? hacc[q] * g.gate_scale : 0.f;
      }
      XChunk hf[2];
      if (XDBG(g, 1024)) {                           // (timing only: the raw accumulator bits as the operand -- no activation, dropout or split)
        hf[0].a = u4v{__float_as_uint(hacc[0]), __float_as_uint(hacc[1]), __float_as_uint(hacc[2]), __float_as_uint(hacc[3])};
        hf[0].b = u4v{__float_as_uint(hacc[4]), __float_as_uint(hacc[5]), __float_as_uint(hacc[6]), __float_as_uint(hacc[7])};
        hf[1].a = u4v{__float_as_uint(hacc[8]), __float_as_uint(hacc[9]), __float_as_uint(hacc[10]), __float_as_uint(hacc[11])};
        hf[1].b = u4v{__float_as_uint(hacc[12]), __float_as_uint(hacc[13]), __float_as_uint(hacc[14]), __float_as_uint(hacc[15])};
      } else {
        bf16x8 hi, lo;
        x3_split8<E>(v, hi, lo);
        hf[0].a = __builtin_bit_cast(u4v, hi); hf[0].b = __builtin_bit_cast(u4v, lo);
        x3_split8<E>(v + 8, hi, lo);
        hf[1].a = __builtin_bit_cast(u4v, hi); hf[1].b = __builtin_bit_cast(u4v, lo);
      }
      const bool st_h = hob != nullptr;                                     // (wave-uniform)
      float* hwave = hob + (blk * 128 + wave * 32) * g.ldh + t * 32;          // row 0 of this wave's strip, this hidden tile's columns
      unsigned short* hwave16 = hob16 + (blk * 128 + wave * 32) * g.ldh + t * 32;
      // ---- second GEMM, K-slice t (u = 0, 1); the hidden tile's stores ride behind the first MFMAs ----
      P.begin_slot();
      if (!XDBG(g, 512))
      x3_slot_tiles<E>(abase + 2 * SLOT_BYTES, abase + 3 * SLOT_BYTES, pre, hf[0], yacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<2>();
        if (i == 4 && st_h) {
          if (HH) tile_store_rows_bf16(stage, v, j, hb, lane, hwave16, g.ldh, wave_ok && !XDBG(g, 16));
          else tile_store_rows(stage, v, j, hb, lane, hwave, g.ldh, wave_ok && !XDBG(g, 16));
        }
      });
      P.begin_slot();
      if (!XDBG(g, 512))
      x3_slot_tiles<E>(abase + 3 * SLOT_BYTES, abase + 0 * SLOT_BYTES, pre, hf[1], yacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<3>();
      });
    }

    // ---------------- final epilogue of the block ----------------
    const uint64_t rowq = ((uint64_t)tok * 256ull) >> 2;
    float* ywave = yb + (blk * 128 + wave * 32) * g.ldy;
    const float* rrow_p = rb + (has_res ? tokc * g.ldr + 16 * hb : 0);
#pragma unroll
    for (int ot = 0; ot < 8; ot++) {
      const int col0 = ot * 32 + 16 * hb;
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = yacc[ot][q];
      if (MODE == 0 && g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_o, rowq + (col0 >> 2), thr, inv_keep);
      if (MODE == 0) {                                // residual = the block input, still in the strip registers (hi + lo)
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
      // these two chunks have fed their last use of the block: their registers take the next block's values
      chunk_load(xr[2 * ot], xrow_next + chunk_off(2 * ot));
      chunk_load(xr[2 * ot + 1], xrow_next + chunk_off(2 * ot + 1));
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) yacc[ot][q] = v[q];
      } else {
        tile_store_rows(stage, v, j, hb, lane, ywave + ot * 32, g.ldy, wave_ok);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 0) {
      const bool p16 = g.flags & HFTT_SL_PRE_BF16;
      x3_ln_rows(yacc, prm_b + p + 256, prm_b + p + 512, hb, g.ln_mean, g.ln_rstd, tok, wave_ok, stage, j, lane,
                 (preb != nullptr && !p16) ? preb + (blk * 128 + wave * 32) * g.ldy : nullptr, ywave, g.ldy,
                 (preb != nullptr && p16) ? reinterpret_cast<unsigned short*>(preb) + (blk * 128 + wave * 32) * g.ldy : nullptr);
    }
  }
  P.drain();
}

// ---------------------------------------------------------------------------------------------------------------------
// fc_o + dropout + residual + LayerNorm  ->  FFN + residual + LayerNorm on ONE strip (hftt_attn_out_ffn_fwd, d = 256, p = 512, fp16 halves):
// the two halves of an encoder / decoder layer behind its attention as one launch.  Phase A is x3_linear_kernel<X3_F16, true, 1, 1, true>'s block
// (16 K-outer slots of fc_o, its epilogue), phase B x3_mlp_kernel<0>'s (64 slots); the LayerNorm output of phase A stays in the accumulator
// registers, is split into the strip's (hi, lo) chunks there and never read back from memory -- in the inference plan it is not even written.
// One weight stream of 80 slots per block: fc_o (order 0), then the FFN's interleaved pair.  Results are bit-identical to the two launches.
// ---------------------------------------------------------------------------------------------------------------------
template <bool HH>
__global__ __launch_bounds__(256, 1) void x3_oln_mlp_kernel(const hftt_strip_desc o, const hftt_ffn_desc g) {
  constexpr int E = X3_F16, PT = 16, p = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const long nblk = ((long)g.M + 127) / 128;
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // bo[256] | gamma1[256] | beta1[256] | b1[512] | b2[256] | gamma2[256] | beta2[256] | store patches
  float* stage = reinterpret_cast<float*>(smem + RING_BYTES + 4 * 2048 + wave * STG_BYTES_PER_WAVE);
  const float* cxb = reinterpret_cast<const float*>(o.x);        // the attention output (phase A's strip)
  float* y1b = reinterpret_cast<float*>(o.C);                    // LayerNorm-1 output: stored for the backward (training plan) or not at all (NULL)
  float* pre1b = reinterpret_cast<float*>(o.pre_ln_out);
  const float* r1b = reinterpret_cast<const float*>(o.residual);
  float* yb = reinterpret_cast<float*>(g.y);
  float* preb = reinterpret_cast<float*>(g.pre_ln_out);
  float* hob = reinterpret_cast<float*>(g.h_out);
  unsigned short* hob16 = reinterpret_cast<unsigned short*>(g.h_out);

  XPipe P;
  P.w = o.w; P.S = 16 + 4 * PT; P.fill_pos = 0;
  P.ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  P.wave = wave; P.lane = lane;
  P.nofill = false; P.nobar = false;

  auto tok_of = [&](long blk) { const long t = blk * 128 + wave * 32 + j; return t < g.M ? t : (long)g.M - 1; };

  prm[tid] = o.bias != nullptr ? o.bias[tid] : 0.f;
  prm[256 + tid] = o.ln_gamma[tid]; prm[512 + tid] = o.ln_beta[tid];
  for (int i = tid; i < p; i += 256) prm[768 + i] = g.b1 != nullptr ? g.b1[i] : 0.f;
  prm[1280 + tid] = g.b2 != nullptr ? g.b2[tid] : 0.f;
  prm[1536 + tid] = g.ln_gamma[tid]; prm[1792 + tid] = g.ln_beta[tid];

  XChunk xr[16];
  {
    const float* p0 = cxb + tok_of(blockIdx.x) * o.ldx + 16 * h;
#pragma unroll
    for (int c = 0; c < 16; c++) chunk_load(xr[c], p0 + chunk_off(c));
  }
  P.fill<0>(); P.fill<1>(); P.fill<2>();
  static_assert(FILL_AHEAD == 3, "prologue fills");
  wait_lgkm0();
  P.prologue_sync();

  const uint32_t thr_a = hftt_keep_thr(o.drop_p), thr = hftt_keep_thr(g.drop_p);
  const float inv_keep_a = hftt_keep_scale(o.drop_p), inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;
    asm volatile("" : "+v"(hb));
    const long tok = blk * 128 + wave * 32 + j;
    const bool wave_ok = (blk * 128 + wave * 32) < g.M;
    const long tokc = tok_of(blk);
    const long nxt = blk + gridDim.x;
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const float* prm_b = prm + zero;

    // ======== phase A: fc_o over the attention-output strip (loaded by the previous block's last epilogue, or the prologue) ========
    chunk_convert<E>(xr[0]);
    {
      f32x16 acc[8];
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
        float b[16];
        lds16f(prm_b + ot * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) acc[ot][q] = b[q];
      }
      XPre pre;
      slot_head(abase, pre);
      static_for<16>([&](auto c_c) __attribute__((always_inline)) {
        constexpr int c = decltype(c_c)::value;
        constexpr int BUF = c & 3;
        P.begin_slot();
        x3_slot_tiles<E>(abase + BUF * SLOT_BYTES, abase + ((BUF + 1) & 3) * SLOT_BYTES, pre, xr[c], acc, [&](auto i_c) __attribute__((always_inline)) {
          constexpr int i = decltype(i_c)::value;
          if (i == 1) P.template refill<BUF>();
          if (i == 9 && c < 15) chunk_convert<E>(xr[c + 1]);
        });
      });
      // ---- epilogue A: scale, dropout (the attention branch's output site), residual = the layer input, LayerNorm 1 ----
      const uint64_t rowq_a = ((uint64_t)tok * 256ull) >> 2;
      const long rrow = o.res_mod > 0 ? (long)((unsigned)tokc % (unsigned)o.res_mod) : tokc;
      const float* rrow_p = r1b + rrow * o.ldr + 16 * hb;
      float rnext[16];
      load16f(rrow_p, rnext);
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
        const int col0 = ot * 32 + 16 * hb;
        float v[16], r[16];
#pragma unroll
        for (int q = 0; q < 16; q++) r[q] = rnext[q];
        if (ot < 7) load16f(rrow_p + (ot + 1) * 32, rnext);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = acc[ot][q] * o.out_scale;
        if (o.drop_p > 0.f) drop16(v, o.drop_seed, o.drop_site, rowq_a + (col0 >> 2), thr_a, inv_keep_a);
#pragma unroll
        for (int q = 0; q < 16; q++) acc[ot][q] = v[q] + r[q];
        __builtin_amdgcn_sched_barrier(0);
      }
      const bool p16a = o.flags & HFTT_SL_PRE_BF16;
      x3_ln_rows<true>(acc, prm_b + 256, prm_b + 512, hb, o.ln_mean, o.ln_rstd, tok, wave_ok, stage, j, lane,
                       (pre1b != nullptr && !p16a) ? pre1b + (blk * 128 + wave * 32) * o.ldc : nullptr,
                       y1b != nullptr ? y1b + (blk * 128 + wave * 32) * o.ldc : nullptr, o.ldc,
                       (pre1b != nullptr && p16a) ? reinterpret_cast<unsigned short*>(pre1b) + (blk * 128 + wave * 32) * o.ldc : nullptr);
      // the normalised rows become the FFN's strip: accumulator register q of tile ot is feature 32 ot + 16 h + q = chunk 2 ot + (q >> 3), element q & 7
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
          xr[2 * ot + u].a = u4v{__float_as_uint(acc[ot][8 * u]), __float_as_uint(acc[ot][8 * u + 1]), __float_as_uint(acc[ot][8 * u + 2]), __float_as_uint(acc[ot][8 * u + 3])};
          xr[2 * ot + u].b = u4v{__float_as_uint(acc[ot][8 * u + 4]), __float_as_uint(acc[ot][8 * u + 5]), __float_as_uint(acc[ot][8 * u + 6]), __float_as_uint(acc[ot][8 * u + 7])};
          chunk_convert<E>(xr[2 * ot + u]);
        }
      }
    }

    // ======== phase B: the FFN block of x3_mlp_kernel<0> on the strip in the registers ========
    const float* xrow_next = cxb + (nxt < nblk ? tok_of(nxt) : tokc) * o.ldx + 16 * hb;
    const uint64_t rowq_h = ((uint64_t)tok * (uint64_t)p) >> 2;
    f32x16 yacc[8];
#pragma unroll
    for (int ot = 0; ot < 8; ot++) {
      float b[16];
      lds16f(prm_b + 1280 + ot * 32 + 16 * hb, b);
#pragma unroll
      for (int q = 0; q < 16; q++) yacc[ot][q] = b[q];
    }
    XPre pre;
    slot_head(abase, pre);
    for (int t = 0; t < PT; t++) {
      f32x16 hacc;
      {
        float b[16];
        lds16f(prm_b + 768 + t * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) hacc[q] = b[q];
      }
      P.begin_slot();
      x3_slot_chunks<E, 0>(abase + 0 * SLOT_BYTES, abase + 1 * SLOT_BYTES, pre, xr, hacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<0>();
      });
      P.begin_slot();
      x3_slot_chunks<E, 8>(abase + 1 * SLOT_BYTES, abase + 2 * SLOT_BYTES, pre, xr, hacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<1>();
      });
      float v[16];
      const int hcol0 = t * 32 + 16 * hb;
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = fmaxf(hacc[q], 0.f);
      if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_h, rowq_h + (hcol0 >> 2), thr, inv_keep);
      XChunk hf[2];
      {
        bf16x8 hi, lo;
        x3_split8<E>(v, hi, lo);
        hf[0].a = __builtin_bit_cast(u4v, hi); hf[0].b = __builtin_bit_cast(u4v, lo);
        x3_split8<E>(v + 8, hi, lo);
        hf[1].a = __builtin_bit_cast(u4v, hi); hf[1].b = __builtin_bit_cast(u4v, lo);
      }
      const bool st_h = hob != nullptr;
      float* hwave = hob + (blk * 128 + wave * 32) * g.ldh + t * 32;
      unsigned short* hwave16 = hob16 + (blk * 128 + wave * 32) * g.ldh + t * 32;
      P.begin_slot();
      x3_slot_tiles<E>(abase + 2 * SLOT_BYTES, abase + 3 * SLOT_BYTES, pre, hf[0], yacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<2>();
        if (i == 4 && st_h) {
          if (HH) tile_store_rows_bf16(stage, v, j, hb, lane, hwave16, g.ldh, wave_ok);
          else tile_store_rows(stage, v, j, hb, lane, hwave, g.ldh, wave_ok);
        }
      });
      P.begin_slot();
      x3_slot_tiles<E>(abase + 3 * SLOT_BYTES, abase + 0 * SLOT_BYTES, pre, hf[1], yacc, [&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if (i == 1) P.template refill<3>();
      });
    }
    // ---- final epilogue: output dropout, residual = the strip (hi + lo), LayerNorm 2; the strip registers take the NEXT block's attention output ----
    const uint64_t rowq = ((uint64_t)tok * 256ull) >> 2;
    float* ywave = yb + (blk * 128 + wave * 32) * g.ldy;
#pragma unroll
    for (int ot = 0; ot < 8; ot++) {
      const int col0 = ot * 32 + 16 * hb;
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = yacc[ot][q];
      if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_o, rowq + (col0 >> 2), thr, inv_keep);
      float r[16];
      chunk_values<E>(xr[2 * ot], r); chunk_values<E>(xr[2 * ot + 1], r + 8);
#pragma unroll
      for (int q = 0; q < 16; q++) yacc[ot][q] = v[q] + r[q];
      chunk_load(xr[2 * ot], xrow_next + chunk_off(2 * ot));
      chunk_load(xr[2 * ot + 1], xrow_next + chunk_off(2 * ot + 1));
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool p16 = g.flags & HFTT_SL_PRE_BF16;
    x3_ln_rows(yacc, prm_b + 1536, prm_b + 1792, hb, g.ln_mean, g.ln_rstd, tok, wave_ok, stage, j, lane,
               (preb != nullptr && !p16) ? preb + (blk * 128 + wave * 32) * g.ldy : nullptr, ywave, g.ldy,
               (preb != nullptr && p16) ? reinterpret_cast<unsigned short*>(preb) + (blk * 128 + wave * 32) * g.ldy : nullptr);
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
template <typename K>
int set_lds(K kernel, int lds, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { hftt_set_error("%s: hipFuncSetAttribute(%d B LDS) failed: %s", what, lds, hipGetErrorString(e)); return 2; }
  return 0;
}
// Resident strip chunks of the one-pass K-outer forms without LayerNorm (the K = 512 / 768 dX products): 8 = half a set, 256 registers (24 bytes
// of scratch on the K = 768 residual form), two workgroups per CU -- one's epilogue under the other's MFMAs.  Same box, three interleaved runs each
// way: 277.07 against 276.17 clips/s (+0.3 %, profiles/r06_ab_dx_two_workgroups.txt).  The LayerNorm form (fc_o + residual + LayerNorm) stays at
// 16: at 256 registers its epilogue spills (620 B per lane with 8 chunks, 468 B with 4, 332 B with the next block's strip loaded only behind
// the epilogue) and the step LOSES 1.1 - 1.2 % in every one of those forms (same file).
#ifndef HFTT_XL_XR
#define HFTT_XL_XR 8
#endif
template <int E, bool LN, int PASSES, int KCH, bool HR>
int launch_xl(const hftt_strip_desc& d, hipStream_t st) {
  constexpr int XR = (PASSES == 1 && !LN) ? HFTT_XL_XR : 16;
  const int lds = RING_BYTES + 4 * (PASSES * 256 + 512) + 4 * STG_BYTES_PER_WAVE;
  static int attr = 0;
  if (lds > attr) { if (int rc = set_lds(x3_linear_kernel<E, LN, PASSES, KCH, HR, XR>, lds, "x3_strip_linear")) return rc; attr = lds; }
  const int cus = n_cus() * (XR == 16 ? 1 : 2);
  if (cus <= 0) { hftt_set_error("x3_strip_linear: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((x3_linear_kernel<E, LN, PASSES, KCH, HR, XR>), dim3((unsigned)(nblk < cus ? nblk : cus)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("x3_strip_linear");
  return 0;
}
template <int E, int NT, bool HR, bool PLN = false, bool XD = false>
int launch_xn(const hftt_strip_desc& d, hipStream_t st) {
  const int lds = RING_BYTES + 4 * d.N + 4 * STG_BYTES_PER_WAVE;
  static int attr = 0;
  if (lds > attr) { if (int rc = set_lds(x3_linear_n_kernel<E, NT, HR, PLN, XD>, lds, "x3_strip_linear")) return rc; attr = lds; }
  const int cus = n_cus();
  if (cus <= 0) { hftt_set_error("x3_strip_linear: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((x3_linear_n_kernel<E, NT, HR, PLN, XD>), dim3((unsigned)(nblk < 2 * cus ? nblk : 2 * cus)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("x3_strip_linear");
  return 0;
}
template <int MODE, bool HH, bool GH = false>
int launch_xm(const hftt_ffn_desc& d, hipStream_t st) {
  const int lds = RING_BYTES + 4 * (d.p + 768) + 4 * STG_BYTES_PER_WAVE;
  static int attr = 0;
  if (lds > attr) { if (int rc = set_lds(x3_mlp_kernel<MODE, 16, HH, GH>, lds, "x3_strip_mlp")) return rc; attr = lds; }
  const int cus = n_cus();
  if (cus <= 0) { hftt_set_error("x3_strip_mlp: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((x3_mlp_kernel<MODE, 16, HH, GH>), dim3((unsigned)(nblk < cus ? nblk : cus)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("x3_strip_mlp");
  return 0;
}

template <bool HH>
int launch_xom(const hftt_strip_desc& o, const hftt_ffn_desc& d, hipStream_t st) {
  const int lds = RING_BYTES + 4 * 2048 + 4 * STG_BYTES_PER_WAVE;
  static int attr = 0;
  if (lds > attr) { if (int rc = set_lds(x3_oln_mlp_kernel<HH>, lds, "x3_attn_out_ffn")) return rc; attr = lds; }
  const int cus = n_cus();
  if (cus <= 0) { hftt_set_error("x3_attn_out_ffn: device query failed"); return 2; }
  const long nblk = ((long)d.M + 127) / 128;
  hipLaunchKernelGGL((x3_oln_mlp_kernel<HH>), dim3((unsigned)(nblk < cus ? nblk : cus)), dim3(256), lds, st, o, d);
  HFTT_CHECK_LAUNCH("x3_attn_out_ffn");
  return 0;
}

#include "x3s_strip.h"

template <int E>
int dispatch_xl(const hftt_strip_desc& d, hipStream_t st) {
  const int passes = d.N / 256, kch = d.K / 256;
  const bool hr = d.residual != nullptr;
#define HFTT_XL(LN_, P_, K_) return hr ? launch_xl<E, LN_, P_, K_, true>(d, st) : launch_xl<E, LN_, P_, K_, false>(d, st)
  if (d.ln_gamma != nullptr) {
    if (kch == 1) HFTT_XL(true, 1, 1);
    if (kch == 2) HFTT_XL(true, 1, 2);
    hftt_set_error("x3_strip_linear: the LayerNorm form covers K = 256 / 512 (got %d)", d.K);
    return 1;
  }
  if constexpr (E == X3_F16) {
    if (d.flags & HFTT_SL_C_F16PAIR) {                // (validated by the caller: no LayerNorm, no residual, K == 256)
      if (passes == 1) return launch_xn<E, 8, false, true>(d, st);
      if (passes == 2) return launch_xn<E, 16, false, true>(d, st);
      if (passes == 3) return launch_xn<E, 24, false, true>(d, st);
      if (passes == 4) return launch_xn<E, 32, false, true>(d, st);      // the cross-attention K / V of two / three decoder layers in one launch
      if (passes == 6) return launch_xn<E, 48, false, true>(d, st);
      hftt_set_error("x3_strip_linear: the plane form covers N = 256, 512, 768, 1024, 1536 (got %d)", d.N);
      return 1;
    }
  }
  if constexpr (E == X3_BF16) {
    if (d.flags & HFTT_SL_X_DROP) return launch_xn<E, 8, false, false, true>(d, st);      // (validated by the caller: N == K == 256, no residual)
  }
  // K == 256 without LayerNorm: the output-tile-major kernel (tile-major pack, two workgroups per CU)
  if (kch == 1 && passes == 1) return hr ? launch_xn<E, 8, true>(d, st) : launch_xn<E, 8, false>(d, st);
  if (kch == 1 && passes == 2) return hr ? launch_xn<E, 16, true>(d, st) : launch_xn<E, 16, false>(d, st);
  if (kch == 1 && passes == 3) return hr ? launch_xn<E, 24, true>(d, st) : launch_xn<E, 24, false>(d, st);
  if (kch == 2 && passes == 1) HFTT_XL(false, 1, 2);
  if (kch == 3 && passes == 1) HFTT_XL(false, 1, 3);
#undef HFTT_XL
  hftt_set_error("x3_strip_linear: shape N=%d K=%d is not covered (N/256 x K/256 in {1x1, 2x1, 3x1, 1x2, 1x3})", d.N, d.K);
  return 1;
}

}  // namespace

// HFTT_X3_DEBUG (timing experiments, results garbage): 1 no ring fills, 2 no slot waits / barriers, 16 no result stores, 32 no strip reloads,
// 64 no strip conversion; fused block: 128 no hidden dropout, 256 no first-GEMM MFMAs, 512 no second-GEMM MFMAs
static int x3_debug() { static const int v = [] { const char* e = getenv("HFTT_X3_DEBUG"); return e ? atoi(e) : 0; }(); return v; }

int hftt_x3_strip_linear(const hftt_strip_desc& d0, hipStream_t st) {
  hftt_strip_desc d = d0;
  d.pad = x3_debug();
  if (d.K <= 192 && d.N <= 192) {                     // the small-width family (weights resident in LDS, compact pack): x3s_strip.h
    HFTT_REQUIRE(d.K % 32 == 0 && d.N % 32 == 0 && d.M % 32 == 0 && d.K > 0 && d.N > 0, "x3s_strip_linear: needs K %% 32 == 0, N %% 32 == 0, M %% 32 == 0 (M=%d N=%d K=%d)", d.M, d.N, d.K);
    HFTT_REQUIRE(d.gate == nullptr && !(d.flags & (HFTT_SL_X_BF16 | HFTT_SL_C_BF16 | HFTT_SL_RES_BF16 | HFTT_SL_X3_GRAD_HI | HFTT_SL_C_F16PAIR)),
                 "x3s_strip_linear: fp32 tensors, no gate / gradient-rounding / plane forms");
    HFTT_REQUIRE(d.ldx % 4 == 0 && d.ldc % 4 == 0 && (d.residual == nullptr || d.ldr % 4 == 0), "x3s_strip_linear: rows must be 16-byte aligned");
    HFTT_REQUIRE(((uintptr_t)d.w & 15) == 0, "x3s_strip_linear: the weight stream must be 16-byte aligned");
    return (d.flags & HFTT_SL_X3_BF16) ? dispatch_xs<X3_BF16>(d, st) : dispatch_xs<X3_F16>(d, st);
  }
  HFTT_REQUIRE(d.K % 256 == 0 && d.N % 256 == 0 && d.M % 32 == 0, "x3_strip_linear: needs K %% 256 == 0, N %% 256 == 0, M %% 32 == 0 (M=%d N=%d K=%d)", d.M, d.N, d.K);
  HFTT_REQUIRE(d.gate == nullptr, "x3_strip_linear: no gate form (the fused block hftt_ffn_bwd_dx carries the gate)");
  HFTT_REQUIRE(!(d.flags & (HFTT_SL_X_BF16 | HFTT_SL_C_BF16 | HFTT_SL_RES_BF16)), "x3_strip_linear: tensors are fp32 in the split modes");
  HFTT_REQUIRE(d.ldx % 4 == 0 && d.ldc % 4 == 0 && (d.residual == nullptr || d.ldr % 4 == 0), "x3_strip_linear: rows must be 16-byte aligned");
  HFTT_REQUIRE(!(d.flags & HFTT_SL_X3_GRAD_HI) || (d.flags & HFTT_SL_X3_BF16), "x3_strip_linear: HFTT_SL_X3_GRAD_HI goes with HFTT_SL_X3_BF16");
  if (d.flags & HFTT_SL_X_DROP) {
    HFTT_REQUIRE((d.flags & HFTT_SL_X3_BF16) && !(d.flags & HFTT_SL_X3_GRAD_HI) && d.K == 256 && d.N == 256 && d.ldx == 256 && d.ln_gamma == nullptr && d.residual == nullptr &&
                 !(d.flags & (HFTT_SL_RELU | HFTT_SL_C_F16PAIR)),
                 "x3_strip_linear: HFTT_SL_X_DROP is the backward of a dropout in front of a 256 -> 256 projection (HFTT_SL_X3_BF16, N == K == ldx == 256, no LayerNorm / residual / ReLU)");
  }
  if (d.flags & HFTT_SL_C_F16PAIR) {
    HFTT_REQUIRE(!(d.flags & HFTT_SL_X3_BF16) && d.K == 256 && d.N <= 1536 && d.ln_gamma == nullptr && d.residual == nullptr && !(d.flags & HFTT_SL_RELU) && !(d.drop_p > 0.f),
                 "x3_strip_linear: HFTT_SL_C_F16PAIR is the plain forward projection (HFTT_SL_X3_F16, K == 256, N <= 1536, no LayerNorm / residual / ReLU / dropout)");
    HFTT_REQUIRE(((uintptr_t)d.C & 15) == 0, "x3_strip_linear: C must be 16-byte aligned");
  }
#ifdef HFTT_GRAD_HI_BUILD
  if (d.flags & HFTT_SL_X3_GRAD_HI) return dispatch_xl<X3_BF16H>(d, st);
#else
  HFTT_REQUIRE(!(d.flags & HFTT_SL_X3_GRAD_HI), "x3_strip_linear: this library was built without the gradient-rounding option (HFTT_BUILD_GRAD_HI=1 python nylon-amt_amd/build.py)");
#endif
  return (d.flags & HFTT_SL_X3_BF16) ? dispatch_xl<X3_BF16>(d, st) : dispatch_xl<X3_F16>(d, st);
}

int hftt_x3_strip_mlp(const hftt_ffn_desc& d0, hipStream_t st) {
  hftt_ffn_desc d = d0;
  d.pad = x3_debug();
  if (d.d == 64 && d.p == 128) {                      // the reference's default width: x3s_strip.h
    HFTT_REQUIRE(d.M % 32 == 0, "x3s_strip_mlp: needs M %% 32 == 0 (M=%d)", d.M);
    HFTT_REQUIRE(!(d.flags & (HFTT_SL_X_BF16 | HFTT_SL_C_BF16 | HFTT_SL_RES_BF16 | HFTT_SL_X3_GRAD_HI)), "x3s_strip_mlp: fp32 tensors, no gradient-rounding form");
    HFTT_REQUIRE(d.mode == 1 || d.residual == nullptr, "x3s_strip_mlp: the forward block's residual is its input");
    HFTT_REQUIRE(((d.flags & HFTT_SL_X3_BF16) != 0) == (d.mode == 1), "x3s_strip_mlp: mode 0 takes fp16 halves (HFTT_SL_X3_F16), mode 1 bf16 halves");
    HFTT_REQUIRE(d.ldx % 4 == 0 && d.ldy % 4 == 0 && ((uintptr_t)d.w & 15) == 0, "x3s_strip_mlp: rows / weight stream must be 16-byte aligned");
    if (d.flags & HFTT_SL_H_BF16) {
      HFTT_REQUIRE((d.h_out == nullptr || d.ldh % 8 == 0) && (d.gate == nullptr || d.ldg % 8 == 0), "x3s_strip_mlp: bf16 hidden rows must be 16-byte aligned");
      return d.mode == 0 ? launch_xsm<0, true>(d, st) : launch_xsm<1, true>(d, st);
    }
    return d.mode == 0 ? launch_xsm<0, false>(d, st) : launch_xsm<1, false>(d, st);
  }
  HFTT_REQUIRE(d.p == 512 && d.M % 32 == 0, "x3_strip_mlp: needs p == 512 and M %% 32 == 0 (M=%d p=%d)", d.M, d.p);
  HFTT_REQUIRE(!(d.flags & (HFTT_SL_X_BF16 | HFTT_SL_C_BF16 | HFTT_SL_RES_BF16)), "x3_strip_mlp: tensors are fp32 in the split modes");
  HFTT_REQUIRE(d.mode == 1 || d.residual == nullptr, "x3_strip_mlp: the forward block's residual is its input");
  HFTT_REQUIRE(((d.flags & HFTT_SL_X3_BF16) != 0) == (d.mode == 1), "x3_strip_mlp: mode 0 takes fp16 halves (HFTT_SL_X3_F16), mode 1 bf16 halves");
  const bool gh = (d.flags & HFTT_SL_X3_GRAD_HI) != 0;
  HFTT_REQUIRE(!gh || d.mode == 1, "x3_strip_mlp: HFTT_SL_X3_GRAD_HI belongs to the backward form");
#ifndef HFTT_GRAD_HI_BUILD
  HFTT_REQUIRE(!gh, "x3_strip_mlp: this library was built without the gradient-rounding option (HFTT_BUILD_GRAD_HI=1 python nylon-amt_amd/build.py)");
#endif
  if (d.flags & HFTT_SL_H_BF16) {
    HFTT_REQUIRE((d.h_out == nullptr || d.ldh % 8 == 0) && (d.gate == nullptr || d.ldg % 8 == 0), "x3_strip_mlp: bf16 hidden rows must be 16-byte aligned");
    if (d.mode == 0) return launch_xm<0, true>(d, st);
#ifdef HFTT_GRAD_HI_BUILD
    if (gh) return launch_xm<1, true, true>(d, st);
#endif
    return launch_xm<1, true, false>(d, st);
  }
  if (d.mode == 0) return launch_xm<0, false>(d, st);
#ifdef HFTT_GRAD_HI_BUILD
  if (gh) return launch_xm<1, false, true>(d, st);
#endif
  return launch_xm<1, false, false>(d, st);
}

// hftt_attn_out_ffn_fwd: the fc_o + LayerNorm descriptor and the FFN descriptor of the two launches it replaces, unchanged, with o.C (the
// LayerNorm-1 output) optional and the two weight streams adjacent (fc_o's 16 slots, then the FFN's 64)
int hftt_x3_attn_out_ffn(const hftt_strip_desc& o, const hftt_ffn_desc& d, hipStream_t st) {
  HFTT_REQUIRE(o.M == d.M && o.M > 0 && o.M % 32 == 0 && o.N == 256 && o.K == 256 && d.d == 256 && d.p == 512 && d.mode == 0,
               "attn_out_ffn_fwd: needs M %% 32 == 0, fc_o 256 -> 256, FFN d = 256, p = 512, mode 0 (M=%d/%d N=%d K=%d d=%d p=%d mode=%d)", o.M, d.M, o.N, o.K, d.d, d.p, d.mode);
  HFTT_REQUIRE((o.flags & HFTT_SL_X3_F16) && (d.flags & HFTT_SL_X3_F16) && !((o.flags | d.flags) & (HFTT_SL_X3_BF16 | HFTT_SL_X_BF16 | HFTT_SL_C_BF16 | HFTT_SL_RES_BF16 | HFTT_SL_RELU | HFTT_SL_C_F16PAIR | HFTT_SL_X_DROP | HFTT_SL_X3_GRAD_HI)),
               "attn_out_ffn_fwd: both descriptors in the forward split mode (HFTT_SL_X3_F16), fp32 tensors");
  HFTT_REQUIRE(o.x != nullptr && o.w != nullptr && o.residual != nullptr && o.gate == nullptr && o.ln_gamma != nullptr && o.ln_beta != nullptr,
               "attn_out_ffn_fwd: the first half is fc_o + residual + LayerNorm (x, w, residual, gamma, beta; no gate)");
  HFTT_REQUIRE(d.w == o.w + 16 * 16 * 512, "attn_out_ffn_fwd: the FFN's weight stream must follow fc_o's 16 slots (one stream of 80 slots per block)");
  HFTT_REQUIRE(d.residual == nullptr && d.y != nullptr && d.ln_gamma != nullptr && d.ln_beta != nullptr && d.ldy == 256, "attn_out_ffn_fwd: FFN half needs y, gamma, beta, ldy == 256 and no residual pointer (the residual is the strip)");
  HFTT_REQUIRE(o.C == nullptr || o.C == d.x, "attn_out_ffn_fwd: the FFN's input IS the first half's output (o.C == f.x, or o.C == NULL when it is not stored)");
  HFTT_REQUIRE(o.ldx % 4 == 0 && o.ldc % 4 == 0 && o.ldr % 4 == 0 && (d.h_out == nullptr || d.ldh % ((d.flags & HFTT_SL_H_BF16) ? 8 : 4) == 0), "attn_out_ffn_fwd: rows must be 16-byte aligned");
  HFTT_REQUIRE((((uintptr_t)o.x | (uintptr_t)o.w | (uintptr_t)o.C | (uintptr_t)o.residual | (uintptr_t)o.pre_ln_out | (uintptr_t)d.y | (uintptr_t)d.h_out | (uintptr_t)d.pre_ln_out) & 15) == 0,
               "attn_out_ffn_fwd: operands must be 16-byte aligned");
  HFTT_REQUIRE(o.drop_p >= 0.f && o.drop_p < 1.f && d.drop_p >= 0.f && d.drop_p < 1.f && (long)d.M * d.p < (1L << 33), "attn_out_ffn_fwd: drop_p / M out of range");
  return (d.flags & HFTT_SL_H_BF16) ? launch_xom<true>(o, d, st) : launch_xom<false>(o, d, st);
}


extern "C" int hftt_x3_strip_pack(const float* params, uint16_t* wstrip, const hftt_strip_pack_entry* table_dev, int n_entries, int elem, void* stream) {
  HFTT_REQUIRE(params != nullptr && wstrip != nullptr && table_dev != nullptr && n_entries > 0, "x3_strip_pack: null argument");
  HFTT_REQUIRE(((uintptr_t)wstrip & 15) == 0 && ((uintptr_t)params & 15) == 0, "x3_strip_pack: buffers must be 16-byte aligned");
  HFTT_REQUIRE(elem == X3_F16 || elem == X3_BF16, "x3_strip_pack: elem must be 2 (fp16 halves) or 4 (bf16 halves)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (elem == X3_F16) hipLaunchKernelGGL(x3_strip_pack_kernel<X3_F16>, dim3(32, (unsigned)n_entries), dim3(256), 0, st, params, wstrip, table_dev);
  else hipLaunchKernelGGL(x3_strip_pack_kernel<X3_BF16>, dim3(32, (unsigned)n_entries), dim3(256), 0, st, params, wstrip, table_dev);
  HFTT_CHECK_LAUNCH("x3_strip_pack");
  return 0;
}
