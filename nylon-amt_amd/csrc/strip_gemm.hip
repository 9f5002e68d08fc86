// "Strip" kernels of the bf16 mode (gfx950): token strips in registers, weights streamed through an LDS-DMA ring.
// See include/hftt_hip.h (hftt_strip_pack / hftt_strip_linear / hftt_ffn_res_ln_fwd / hftt_ffn_bwd_dx) for the contract.
//
// Geometry.  Workgroup = 4 waves = 128 tokens; wave w owns tokens [128*b + 32*w, +32).  Lane (j = lane & 31, h = lane >> 5)
// belongs to token j of the strip and owns, of every aligned group of 32 features, the 16 features [16h, 16h + 16).
//   * B operand (activations): for feature group pt the lane holds x[token][32pt + 16h + 0..15] = two bf16x8 registers quads
//     (k-steps u = 0, 1) -- one contiguous 32-byte read per group.
//   * A operand (weights): 1 KB fragments in "strip pack" order (hftt_strip_pack), lane-linear, so one LDS-DMA instruction
//     (global_load_lds_dwordx4: wave-uniform LDS base + 16 B per lane) lands a whole fragment and one conflict-free
//     ds_read_b128 per MFMA fetches it.  Rows of a fragment are permuted (c(i)) so that accumulator register g of the lane is
//     feature 16h + g of the tile: C^T comes out in exactly the layout the B operand has.
//   * LDS: ring of 4 slots x 16 KB (16 fragments = 16 MFMAs per wave per slot), one s_barrier per slot, fills issued three
//     slots ahead and retired with counted s_waitcnt vmcnt (the DMA is issued from inline asm: hipcc would otherwise order
//     every ds_read behind the youngest DMA).  Two workgroups per CU run free of each other, so one's epilogue (VALU,
//     stores) overlaps the other's MFMA phase.
#include <stdlib.h>
#include "hftt_common.h"
#include "hftt_host.h"
#include "strip_internal.h"
#include "x3_internal.h"
#include "../../include/hftt_hip.h"

namespace {

constexpr int SLOT_BYTES = 16384;
constexpr int NSLOT = 4;
constexpr int RING_BYTES = NSLOT * SLOT_BYTES;
constexpr int FILL_AHEAD = NSLOT - 1;

// Ablation build (tools/ablate_strip.sh compiles this file with -DHFTT_STRIP_ABLATE into its own library): the descriptor's pad
// field switches single mechanisms off so their cost can be read from the timing difference (results are then garbage).
//   1 no LDS-DMA fills / waits in the main loop   2 no fragment reads + MFMAs   4 no epilogue stores   8 no barriers
#ifdef HFTT_STRIP_ABLATE
#define ABL(g, bit) (((g).pad & (bit)) != 0)
// bit 64: lane 0 of every wave writes shader-clock stamps to the (otherwise unused) ln_mean buffer: [wave id][16] uint64
#define STAMP(g, k) do { if (ABL(g, 64) && lane == 0) reinterpret_cast<unsigned long long*>((g).ln_mean)[((long)blockIdx.x * 4 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ABL(g, bit) false
#define STAMP(g, k) do { } while (0)
#endif

// MFMA row i of a weight fragment <-> feature c(i) of the 32-wide tile; finv is the inverse
__host__ __device__ inline int strip_c_of_i(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
__host__ __device__ inline int strip_i_of_c(int c) { const int g = c & 15, hh = c >> 4; return (g & 3) + 8 * (g >> 2) + 4 * hh; }

// ---------------------------------------------------------------------------------------------------------------------
// weight packing: one thread per 16-byte destination chunk (8 consecutive k of one logical row)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void strip_pack_kernel(const float* __restrict__ params, unsigned short* __restrict__ dst,
                                                         const hftt_strip_pack_entry* __restrict__ table) {
  const hftt_strip_pack_entry e = table[blockIdx.y];
  const int nrows = e.transpose ? e.cols : e.rows;      // logical rows (n) covered by this entry
  const int nk = e.transpose ? e.rows : e.cols;         // logical k covered
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
    const int tile = n >> 5, i = strip_i_of_c(n & 31);
    const int pt = k >> 5, hk = (k >> 4) & 1, u = (k >> 3) & 1;
    long slot, frag;
    if (e.order == 0) { slot = (long)(tile >> 3) * (e.K >> 5) + pt; frag = u * 8 + (tile & 7); }
    else { slot = tile; frag = pt * 2 + u; }
    const long off = ((e.slot_offset + (long)e.slot_stride * slot) * 16 + frag) * 512 + (hk * 32 + i) * 8;
    uint4 o;
    o.x = f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); o.y = f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    o.z = f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16); o.w = f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
    *reinterpret_cast<uint4*>(dst + e.dst_off + off) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA of 16 B per lane: LDS destination = wave-uniform byte address `lds_dst` + 16*lane; source = per-lane pointer.
// M0 is compiler-reserved: save / set / restore inside the one statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// slot s of S has landed once at most the DMAs of the (up to two) younger slots already issued are outstanding
__device__ __forceinline__ void wait_slot(int s, int S) {
  if (s + 2 < S) wait_vmcnt<8>();
  else if (s + 1 < S) wait_vmcnt<4>();
  else wait_vmcnt<0>();
}
// the same right after an epilogue that issued at least NST vector stores behind this slot's DMA (vmcnt retires in issue order: ops
// younger than the awaited DMA may all stay in flight, so the previous pass's stores drain under this pass's MFMAs).  NST must be a
// LOWER bound of the stores the compiler emitted: too small only waits longer, too large would read the slot early.
template <int NST>
__device__ __forceinline__ void wait_slot_after_stores(int s, int S) {
  if (s + 2 < S) wait_vmcnt<8 + NST>();
  else if (s + 1 < S) wait_vmcnt<4 + NST>();
  else wait_vmcnt<NST>();
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2_t));
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  uint4 o;
  o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
  return __builtin_bit_cast(bf16x8, o);
}
__device__ __forceinline__ void unpack8(uint4 q, float* v) {
  v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xFFFF0000u);
  v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xFFFF0000u);
  v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xFFFF0000u);
  v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xFFFF0000u);
}
// 16 consecutive activations (fp32 or bf16 storage) at element offset `off`
__device__ __forceinline__ void load16(const void* base, bool bf, long off, float* v) {
  if (bf) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + off);
    const uint4 a = p[0], b = p[1];
    unpack8(a, v); unpack8(b, v + 8);
  } else {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
#pragma unroll
    for (int q = 0; q < 4; q++) { const float4 t = p[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
  }
}
__device__ __forceinline__ void store16(void* base, bool bf, long off, const float* v) {
  if (bf) {
    uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + off);
    uint4 a, b;
    a.x = pack2(v[0], v[1]); a.y = pack2(v[2], v[3]); a.z = pack2(v[4], v[5]); a.w = pack2(v[6], v[7]);
    b.x = pack2(v[8], v[9]); b.y = pack2(v[10], v[11]); b.z = pack2(v[12], v[13]); b.w = pack2(v[14], v[15]);
    p[0] = a; p[1] = b;
  } else {
    float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off);
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  }
}
// 16 fp32 parameters from LDS (all lanes of a half read the same address: broadcast)
__device__ __forceinline__ void lds16f(const float* p, float* v) {
#pragma unroll
  for (int q = 0; q < 4; q++) { const float4 t = reinterpret_cast<const float4*>(p)[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}
// dropout on 16 consecutive elements whose first element index is 2*q0 (even): one hash per pair
__device__ __forceinline__ void drop16(float* v, uint64_t seed, uint32_t site, uint64_t q0, uint32_t thr, float inv_keep) {
  // q0 = (index of the first element) / 4: the 16 elements are four hash quads (hftt_keep: byte idx&3 of hash(idx>>2) < thr)
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const uint32_t w = hftt_hash(seed, site, q0 + e);
    // byte < thr as an arithmetic mask (sign of byte - thr).  A compare + select per element parks one SGPR pair per decision and hipcc
    // hoists all 64 pairs of an epilogue: the scalar file spilled (231 SGPRs).  The shift is inline asm because instcombine turns
    // (x - thr) >> 31 back into that compare.
    uint32_t m[4];
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m[0]) : "v"((w & 0xFFu) - thr));
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m[1]) : "v"(((w >> 8) & 0xFFu) - thr));
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m[2]) : "v"(((w >> 16) & 0xFFu) - thr));
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m[3]) : "v"((w >> 24) - thr));
#pragma unroll
    for (int f = 0; f < 4; f++) v[4 * e + f] = __uint_as_float(__float_as_uint(v[4 * e + f] * inv_keep) & m[f]);
  }
}

// stream slot s -> ring buffer s % 4 (nothing past the last slot: the tail waits below count exactly what is in flight);
// wave w moves fragments 4w .. 4w+3 of the slot
__device__ __forceinline__ void ring_fill(const unsigned short* w, int s, int s_last, unsigned ring, int wave, int lane) {
  if (s > s_last) return;                           // wave-uniform
  const unsigned short* src = w + ((long)s * 16 + wave * 4) * 512 + lane * 8;
  const unsigned dst = ring + (unsigned)(s & (NSLOT - 1)) * SLOT_BYTES + (unsigned)wave * 4096u;
#pragma unroll
  for (int i = 0; i < 4; i++) glds16(src + i * 512, dst + i * 1024);
}

// activation fragments of one 256-feature chunk: xf[pt][u], pt < KPC
template <bool XBF>
__device__ __forceinline__ void load_xfrags(bf16x8 (&xf)[8][2], const void* x, long off, int kpc) {
#pragma unroll
  for (int pt = 0; pt < 8; pt++) {
    if (pt < kpc) {
      if (XBF) {
        const bf16x8* p = reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned short*>(x) + off + pt * 32);
        xf[pt][0] = p[0]; xf[pt][1] = p[1];
      } else {
        float v[16];
        load16(x, false, off + pt * 32, v);
        xf[pt][0] = pack8(v); xf[pt][1] = pack8(v + 8);
      }
    }
  }
}

// LayerNorm epilogue over the 256 features of the lane's token: the lane holds 128 of them (acc[ot][g]), its partner (lane ^ 32)
// the other 128.  v = acc (bias already inside) -> dropout -> + residual -> LN.
struct LnArgs {
  const float* gamma_lds; const float* beta_lds;
  void* pre_ln_out; float* mean; float* rstd; void* y; long ldy; bool out_bf;
};
__device__ __forceinline__ void ln_epilogue(f32x16 (&acc)[8], const LnArgs& a, long tok, bool tok_ok, int h) {
  float s = 0.f;
#pragma unroll
  for (int ot = 0; ot < 8; ot++)
#pragma unroll
    for (int g = 0; g < 16; g++) s += acc[ot][g];
  const float mean = xor32_sum(s) * (1.0f / 256.0f);
  float q = 0.f;
#pragma unroll
  for (int ot = 0; ot < 8; ot++)
#pragma unroll
    for (int g = 0; g < 16; g++) { const float dlt = acc[ot][g] - mean; q += dlt * dlt; }
  const float rstd = 1.0f / sqrtf(xor32_sum(q) * (1.0f / 256.0f) + 1e-5f);
  if (tok_ok) {
    if (h == 0) {
      if (a.mean != nullptr) a.mean[tok] = mean;
      if (a.rstd != nullptr) a.rstd[tok] = rstd;
    }
#pragma unroll
    for (int ot = 0; ot < 8; ot++) {
      float v[16], ga[16], be[16];
#pragma unroll
      for (int g = 0; g < 16; g++) v[g] = acc[ot][g];
      if (a.pre_ln_out != nullptr) store16(a.pre_ln_out, a.out_bf, tok * a.ldy + ot * 32 + 16 * h, v);
      lds16f(a.gamma_lds + ot * 32 + 16 * h, ga);
      lds16f(a.beta_lds + ot * 32 + 16 * h, be);
#pragma unroll
      for (int g = 0; g < 16; g++) v[g] = (v[g] - mean) * rstd * ga[g] + be[g];
      store16(a.y, a.out_bf, tok * a.ldy + ot * 32 + 16 * h, v);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// C = epi(x . Wl^T + bias), N = 256 * passes
// ---------------------------------------------------------------------------------------------------------------------
template <bool XBF, bool CBF, bool LN>
__global__ __launch_bounds__(256, 2) void strip_linear_kernel(const hftt_strip_desc g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const long tok = (long)blockIdx.x * 128 + wave * 32 + j;
  const bool tok_ok = tok < g.M;
  const long tokc = tok_ok ? tok : (long)g.M - 1;
  const int KP = g.K >> 5;
  const int KCH = (KP + 7) >> 3;                    // 256-feature chunks of K
  const int kpc = KP < 8 ? KP : 8;                  // k pairs per chunk
  const int passes = LN ? 1 : (g.N >> 8);          // (with LayerNorm N == 256; a real loop lets LICM hoist the 128 dropout masks above the k loop: spills)
  const int S = passes * KP;
  const unsigned ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // bias[N] | gamma[256] | beta[256]
  constexpr bool c_bf = CBF;
  const bool res_bf = LN ? true : (bool)(g.flags & HFTT_SL_RES_BF16);       // the LayerNorm form is all-bf16 (host check)
  const long rrow = g.res_mod > 0 ? (long)((unsigned)tokc % (unsigned)g.res_mod) : tokc;

#ifdef HFTT_STRIP_ABLATE
  if (blockIdx.x >= 256 && blockIdx.x < 512 && (g.pad >> 8) > 0) {        // stagger experiment: the second workgroup of each CU starts late (first generation only)
    for (int i = 0; i < (g.pad >> 8); i++) __builtin_amdgcn_s_sleep(16);      // ~1024 cycles per iteration
  }
#endif
  STAMP(g, 0);
  bf16x8 xf[8][2];
#ifdef HFTT_STRIP_ABLATE
  if (ABL(g, 16)) {      // same bytes per wave, but every wave instruction reads 1 KB contiguous (wrong data: timing only)
    const long wb = ((long)blockIdx.x * 128 + wave * 32) * g.ldx;
#pragma unroll
    for (int pt = 0; pt < 8; pt++) {
      const bf16x8* p = reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned short*>(g.x) + wb + pt * 1024 + lane * 8);
      xf[pt][0] = p[0]; xf[pt][1] = p[64];
    }
  } else
#endif
  load_xfrags<XBF>(xf, g.x, tokc * g.ldx + 16 * h, kpc);
  for (int i = tid; i < g.N; i += 256) prm[i] = g.bias != nullptr ? g.bias[i] : 0.f;
  if (LN) {
    prm[g.N + tid] = g.ln_gamma[tid];
    prm[g.N + 256 + tid] = g.ln_beta[tid];
  }
#pragma unroll
  for (int s = 0; s < FILL_AHEAD; s++) ring_fill(g.w, s, S - 1, ring, wave, lane);

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;

  int s = 0;
  STAMP(g, 1);
  for (int pass = 0; pass < passes; pass++) {
    const int n0 = pass * 256;
    f32x16 acc[8];
    for (int kc = 0; kc < KCH; kc++) {
      if (KCH > 1 && (kc > 0 || pass > 0)) load_xfrags<XBF>(xf, g.x, tokc * g.ldx + kc * 256 + 16 * h, kpc);
#pragma unroll
      for (int pt = 0; pt < 8; pt++) {
        if (pt < kpc) {
          if (!ABL(g, 1)) {
            // first three slots of a later pass: the 16 (bf16 C: two dwordx4 per tile) or 32 (fp32 C) stores of the previous pass's
            // epilogue are younger than the slot's DMA (plain epilogue only: gate / residual loads would sit between them)
            if (!LN && pass > 0 && kc == 0 && pt < FILL_AHEAD && g.gate == nullptr && g.residual == nullptr) wait_slot_after_stores<CBF ? 16 : 32>(s, S);
            else wait_slot(s, S);
          }
          if (s == 0) wait_lgkm0();                    // the parameter rows written to LDS above
          if (!ABL(g, 8)) __builtin_amdgcn_s_barrier();
          if (s == 0) STAMP(g, 8);
          if (!ABL(g, 1)) ring_fill(g.w, s + FILL_AHEAD, S - 1, ring, wave, lane);
          if (kc == 0 && pt == 0) {                    // accumulators start from the bias
#pragma unroll
            for (int ot = 0; ot < 8; ot++) {
              float b[16];
              lds16f(prm + n0 + ot * 32 + 16 * h, b);
#pragma unroll
              for (int q = 0; q < 16; q++) acc[ot][q] = b[q];
            }
          }
          const unsigned char* slot = abase + (pt & (NSLOT - 1)) * SLOT_BYTES;
          if (!ABL(g, 2)) {
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int ot = 0; ot < 8; ot++) {
              const bf16x8 a = *reinterpret_cast<const bf16x8*>(slot + (u * 8 + ot) * 1024);
              acc[ot] = mfma32(a, xf[pt][u], acc[ot]);
            }
          }
          s++;
        }
      }
    }
    // ---------------- epilogue of this pass ----------------
    if (!LN) { if (pass == 0) STAMP(g, 2); else if (pass == 1) STAMP(g, 4); else STAMP(g, 6); }
    const bool relu = g.flags & HFTT_SL_RELU;
    const uint64_t rowq = ((uint64_t)tok * (uint64_t)g.N) >> 2;        // quad index of (row, col 0); N % 4 == 0
    if (LN) {
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
        const int col0 = ot * 32 + 16 * h;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = acc[ot][q] * g.out_scale;
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.drop_site, rowq + (col0 >> 2), thr, inv_keep);
        if (g.residual != nullptr) {
          float r[16];
          load16(g.residual, res_bf, rrow * g.ldr + col0, r);
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] += r[q];
        }
#pragma unroll
        for (int q = 0; q < 16; q++) acc[ot][q] = v[q];
        __builtin_amdgcn_sched_barrier(0);             // one tile at a time: hoisting all 64 hashes of the row spills
      }
      LnArgs la{prm + g.N, prm + g.N + 256, g.pre_ln_out, g.ln_mean, g.ln_rstd, g.C, (long)g.ldc, c_bf};
      ln_epilogue(acc, la, tok, tok_ok && !ABL(g, 4), h);
    } else {
#pragma unroll
      for (int ot = 0; ot < 8; ot++) {
        const int col0 = n0 + ot * 32 + 16 * h;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
          float t = acc[ot][q];
          if (relu) t = fmaxf(t, 0.f);
          v[q] = t * g.out_scale;
        }
        if (g.gate != nullptr) {
          float gt[16];
          load16(g.gate, true, tokc * g.ldg + col0, gt);
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] = gt[q] > 0.f ? v[q] * g.gate_scale : 0.f;
        }
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.drop_site, rowq + (col0 >> 2), thr, inv_keep);
        if (g.residual != nullptr) {
          float r[16];
          load16(g.residual, res_bf, rrow * g.ldr + col0, r);
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] += r[q];
        }
#ifdef HFTT_STRIP_ABLATE
        if (ABL(g, 32)) {    // same bytes, 1 KB contiguous per wave instruction (wrong placement: timing only)
          const long wb = ((long)blockIdx.x * 128 + wave * 32) * g.ldc + (long)pass * 256 * 32;
          store16(g.C, c_bf, wb + ot * 1024 + lane * 16, v);
        } else
#endif
        if (tok_ok && !ABL(g, 4)) store16(g.C, c_bf, tok * g.ldc + col0, v);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (pass == 0) STAMP(g, 3); else if (pass == 1) STAMP(g, 5); else STAMP(g, 7);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fused two-GEMM block (d = 256): mode 0 = FFN forward + residual + LayerNorm, mode 1 = dX half of its backward
// ---------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, 2) void strip_mlp_kernel(const hftt_ffn_desc g) {
  constexpr bool XBF = true;                        // the fused block is all-bf16: x, residual, y, pre_ln_out, h (host check)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const long tok = (long)blockIdx.x * 128 + wave * 32 + j;
  const bool tok_ok = tok < g.M;
  const long tokc = tok_ok ? tok : (long)g.M - 1;
  const int PT = g.p >> 5;                          // hidden tiles (even)
  const int S = 2 * PT;
  const unsigned ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // b1[p] | b2[256] | gamma[256] | beta[256]
  constexpr bool c_bf = true, res_bf = true;

#ifdef HFTT_STRIP_ABLATE
  if (blockIdx.x >= 256 && blockIdx.x < 512 && (g.pad >> 8) > 0) {
    for (int i = 0; i < (g.pad >> 8); i++) __builtin_amdgcn_s_sleep(16);
  }
#endif
  bf16x8 xf[8][2];
  load_xfrags<XBF>(xf, g.x, tokc * g.ldx + 16 * h, 8);
  for (int i = tid; i < g.p; i += 256) prm[i] = (MODE == 0 && g.b1 != nullptr) ? g.b1[i] : 0.f;
  prm[g.p + tid] = (MODE == 0 && g.b2 != nullptr) ? g.b2[tid] : 0.f;
  if (MODE == 0) {
    prm[g.p + 256 + tid] = g.ln_gamma[tid];
    prm[g.p + 512 + tid] = g.ln_beta[tid];
  }
#pragma unroll
  for (int s = 0; s < FILL_AHEAD; s++) ring_fill(g.w, s, S - 1, ring, wave, lane);

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  const uint64_t rowq_h = ((uint64_t)tok * (uint64_t)g.p) >> 2;

  f32x16 yacc[8];
  uint4 gnext[2];                                    // mode 1: stored hidden of the next tile (the ReLU / dropout gate)
  if (MODE == 1) {
    const uint4* gp = reinterpret_cast<const uint4*>(g.gate + tokc * g.ldg + 16 * h);
    gnext[0] = gp[0]; gnext[1] = gp[1];
  }

  for (int t0 = 0; t0 < PT; t0 += 2) {
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      const int t = t0 + tt;
      const int sA = 2 * t, sB = 2 * t + 1;
      // ---- first GEMM, hidden tile t: 16 k-steps over the resident strip ----
      if (!ABL(g, 1)) wait_slot(sA, S);
      if (t == 0) wait_lgkm0();
      if (!ABL(g, 8)) __builtin_amdgcn_s_barrier();
      if (!ABL(g, 1)) ring_fill(g.w, sA + FILL_AHEAD, S - 1, ring, wave, lane);
      if (t == 0) {
#pragma unroll
        for (int ot = 0; ot < 8; ot++) {
          float b[16];
          lds16f(prm + g.p + ot * 32 + 16 * h, b);
#pragma unroll
          for (int q = 0; q < 16; q++) yacc[ot][q] = b[q];
        }
      }
      uint4 gcur[2];
      if (MODE == 1) {
        gcur[0] = gnext[0]; gcur[1] = gnext[1];
        const int tn = t + 1 < PT ? t + 1 : t;
        const uint4* gp = reinterpret_cast<const uint4*>(g.gate + tokc * g.ldg + tn * 32 + 16 * h);
        gnext[0] = gp[0]; gnext[1] = gp[1];
      }
      f32x16 hacc;
      {
        float b[16];
        lds16f(prm + t * 32 + 16 * h, b);
#pragma unroll
        for (int q = 0; q < 16; q++) hacc[q] = b[q];
      }
      if (!ABL(g, 2)) {
        const unsigned char* slot = abase + (2 * tt) * SLOT_BYTES;
#pragma unroll
        for (int pt = 0; pt < 8; pt++)
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(slot + (pt * 2 + u) * 1024);
            hacc = mfma32(a, xf[pt][u], hacc);
          }
      }
      // ---- middle epilogue: the lane's 16 hidden features of tile t become the B operand of the second GEMM ----
      float v[16];
      const int hcol0 = t * 32 + 16 * h;
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = fmaxf(hacc[q], 0.f);
        if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_h, rowq_h + (hcol0 >> 2), thr, inv_keep);
      } else {
        float gt[16];
        unpack8(gcur[0], gt); unpack8(gcur[1], gt + 8);
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = gt[q] > 0.f ? hacc[q] * g.gate_scale : 0.f;
      }
      bf16x8 hf[2];
      hf[0] = pack8(v); hf[1] = pack8(v + 8);
      if (g.h_out != nullptr && tok_ok && !ABL(g, 4)) {
        bf16x8* hp = reinterpret_cast<bf16x8*>(g.h_out + tok * g.ldh + hcol0);
        hp[0] = hf[0]; hp[1] = hf[1];
      }
      // ---- second GEMM, K-slice t ----
      if (!ABL(g, 1)) wait_slot(sB, S);
      if (!ABL(g, 8)) __builtin_amdgcn_s_barrier();
      if (!ABL(g, 1)) ring_fill(g.w, sB + FILL_AHEAD, S - 1, ring, wave, lane);
      if (!ABL(g, 2)) {
        const unsigned char* slot = abase + (2 * tt + 1) * SLOT_BYTES;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int ot = 0; ot < 8; ot++) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(slot + (u * 8 + ot) * 1024);
            yacc[ot] = mfma32(a, hf[u], yacc[ot]);
          }
      }
    }
  }

  // ---------------- final epilogue ----------------
  const uint64_t rowq = ((uint64_t)tok * 256ull) >> 2;
#pragma unroll
  for (int ot = 0; ot < 8; ot++) {
    const int col0 = ot * 32 + 16 * h;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = yacc[ot][q];
    if (MODE == 0 && g.drop_p > 0.f) drop16(v, g.drop_seed, g.site_o, rowq + (col0 >> 2), thr, inv_keep);
    if (g.residual != nullptr) {
      float r[16];
      load16(g.residual, res_bf, tokc * g.ldr + col0, r);
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] += r[q];
    } else if (MODE == 0) {                           // residual = the block input, already in registers (bf16)
      float r[16];
      unpack8(__builtin_bit_cast(uint4, xf[ot][0]), r);
      unpack8(__builtin_bit_cast(uint4, xf[ot][1]), r + 8);
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] += r[q];
    }
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 16; q++) yacc[ot][q] = v[q];
    } else if (tok_ok && !ABL(g, 4)) {
      store16(g.y, c_bf, tok * g.ldy + col0, v);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (MODE == 0) {
    LnArgs la{prm + g.p + 256, prm + g.p + 512, g.pre_ln_out, g.ln_mean, g.ln_rstd, g.y, (long)g.ldy, c_bf};
    ln_epilogue(yacc, la, tok, tok_ok && !ABL(g, 4), h);
  }
}

template <typename K>
int set_lds(K kernel, int lds, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { hftt_set_error("%s: hipFuncSetAttribute(%d B LDS) failed: %s", what, lds, hipGetErrorString(e)); return 2; }
  return 0;
}

#ifdef HFTT_STRIP_ABLATE
int ablate_bits() { const char* e = getenv("HFTT_STRIP_ABLATE"); return e ? atoi(e) : 0; }
#endif

template <bool XBF, bool CBF, bool LN>
int launch_linear(const hftt_strip_desc& d0, hipStream_t st) {
  hftt_strip_desc d = d0;
#ifdef HFTT_STRIP_ABLATE
  d.pad = ablate_bits();
#endif
  const int lds = RING_BYTES + 4 * (d.N + 512);
  static int attr = 0;
  if (lds > attr) { if (int rc = set_lds(strip_linear_kernel<XBF, CBF, LN>, lds, "strip_linear")) return rc; attr = lds; }
  hipLaunchKernelGGL((strip_linear_kernel<XBF, CBF, LN>), dim3((unsigned)((d.M + 127) / 128)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("strip_linear");
  return 0;
}
template <int MODE>
int launch_mlp(const hftt_ffn_desc& d0, hipStream_t st) {
  hftt_ffn_desc d = d0;
#ifdef HFTT_STRIP_ABLATE
  d.pad = ablate_bits();
#endif
  const int lds = RING_BYTES + 4 * (d.p + 768);
  static int attr = 0;
  if (lds > attr) { if (int rc = set_lds(strip_mlp_kernel<MODE>, lds, "strip_mlp")) return rc; attr = lds; }
  hipLaunchKernelGGL((strip_mlp_kernel<MODE>), dim3((unsigned)((d.M + 127) / 128)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("strip_mlp");
  return 0;
}

int check_ffn(const hftt_ffn_desc* d, int mode, const char* what) {
  HFTT_REQUIRE(d != nullptr, "%s: null descriptor", what);
  HFTT_REQUIRE(d->mode == mode, "%s: descriptor mode %d", what, d->mode);
  HFTT_REQUIRE(d->M > 0 && d->d == 256, "%s: needs d == 256 (got M=%d d=%d)", what, d->M, d->d);
  HFTT_REQUIRE(d->p >= 64 && d->p % 64 == 0 && d->p <= 4096, "%s: p=%d must be a multiple of 64", what, d->p);
  HFTT_REQUIRE((long)d->M * d->p < (1L << 33), "%s: M*p too large for the 32-bit dropout pair index", what);
  HFTT_REQUIRE(d->x != nullptr && d->w != nullptr && d->y != nullptr, "%s: null operand", what);
  const uint32_t all = HFTT_SL_X_BF16 | HFTT_SL_C_BF16 | HFTT_SL_RES_BF16;
  HFTT_REQUIRE((d->flags & all) == all, "%s: the fused block is all-bf16 (flags X_BF16 | C_BF16 | RES_BF16)", what);
  HFTT_REQUIRE(d->ldx % 8 == 0 && ((uintptr_t)d->x & 15) == 0, "%s: x must be 16-byte aligned per row", what);
  HFTT_REQUIRE(d->ldy % 8 == 0 && ((uintptr_t)d->y & 15) == 0, "%s: y must be 16-byte aligned per row", what);
  HFTT_REQUIRE(((uintptr_t)d->w & 15) == 0, "%s: w must be 16-byte aligned", what);
  HFTT_REQUIRE(d->h_out == nullptr || (d->ldh % 8 == 0 && ((uintptr_t)d->h_out & 15) == 0), "%s: h_out alignment", what);
  HFTT_REQUIRE(d->residual == nullptr || (d->ldr % 8 == 0 && ((uintptr_t)d->residual & 15) == 0), "%s: residual alignment", what);
  HFTT_REQUIRE(d->pre_ln_out == nullptr || ((uintptr_t)d->pre_ln_out & 15) == 0, "%s: pre_ln_out alignment", what);
  HFTT_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "%s: drop_p out of range", what);
  return 0;
}

}  // namespace

extern "C" int hftt_strip_pack(const float* params, uint16_t* wstrip, const hftt_strip_pack_entry* table_dev, int n_entries, void* stream) {
  HFTT_REQUIRE(params != nullptr && wstrip != nullptr && table_dev != nullptr && n_entries > 0, "strip_pack: null argument");
  HFTT_REQUIRE(((uintptr_t)wstrip & 15) == 0 && ((uintptr_t)params & 15) == 0, "strip_pack: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(strip_pack_kernel, dim3(32, (unsigned)n_entries), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params, wstrip, table_dev);
  HFTT_CHECK_LAUNCH("strip_pack");
  return 0;
}

extern "C" int hftt_strip_linear(const hftt_strip_desc* d, void* stream) {
  HFTT_REQUIRE(d != nullptr, "strip_linear: null descriptor");
  if (d->flags & (HFTT_SL_X3_F16 | HFTT_SL_X3_BF16)) {      // split-operand forms (x3_strip.hip): fp32 tensors
    HFTT_REQUIRE(d->M > 0 && d->x != nullptr && d->w != nullptr && d->C != nullptr, "strip_linear: null operand");
    HFTT_REQUIRE((((uintptr_t)d->x | (uintptr_t)d->C | (uintptr_t)d->w | (uintptr_t)d->residual | (uintptr_t)d->pre_ln_out) & 15) == 0, "strip_linear: operands must be 16-byte aligned");
    HFTT_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f && d->res_mod >= 0, "strip_linear: drop_p / res_mod out of range");
    HFTT_REQUIRE((long)d->M * d->N < (1L << 33), "strip_linear: M*N too large for the 32-bit dropout pair index");
    HFTT_REQUIRE(d->ln_gamma == nullptr || ((d->N == 256 || d->N == 64) && d->ln_beta != nullptr && d->ldc == d->N && !(d->flags & HFTT_SL_RELU)),
                 "strip_linear: LayerNorm needs N == ldc == 256 (or 64: the small-width family), beta, no ReLU");
    return hftt_x3_strip_linear(*d, reinterpret_cast<hipStream_t>(stream));
  }
  {                                                   // the bf16 small-width family (bs_strip.hip): K, N <= 192, all-bf16 storage
    const int rc = hftt_bs_strip_linear_try(*d, reinterpret_cast<hipStream_t>(stream));
    if (rc >= 0) return rc;
  }
  HFTT_REQUIRE(d->M > 0 && d->N >= 256 && d->N % 256 == 0 && d->N <= 2048, "strip_linear: N=%d must be a multiple of 256", d->N);
  HFTT_REQUIRE(d->K >= 128 && d->K % 128 == 0 && (d->K <= 256 || d->K % 256 == 0), "strip_linear: K=%d must be 128, 256 or a multiple of 256", d->K);
  HFTT_REQUIRE((long)d->M * d->N < (1L << 33), "strip_linear: M*N too large for the 32-bit dropout pair index");
  HFTT_REQUIRE(d->x != nullptr && d->w != nullptr && d->C != nullptr, "strip_linear: null operand");
  const int xa = (d->flags & HFTT_SL_X_BF16) ? 8 : 4, ca = (d->flags & HFTT_SL_C_BF16) ? 8 : 4;
  HFTT_REQUIRE(d->ldx % xa == 0 && ((uintptr_t)d->x & 15) == 0, "strip_linear: x must be 16-byte aligned per row");
  HFTT_REQUIRE(d->ldc % ca == 0 && ((uintptr_t)d->C & 15) == 0, "strip_linear: C must be 16-byte aligned per row");
  HFTT_REQUIRE(((uintptr_t)d->w & 15) == 0, "strip_linear: w must be 16-byte aligned");
  HFTT_REQUIRE(d->gate == nullptr || (d->ldg % 8 == 0 && ((uintptr_t)d->gate & 15) == 0), "strip_linear: gate alignment");
  HFTT_REQUIRE(d->residual == nullptr || (d->ldr % ((d->flags & HFTT_SL_RES_BF16) ? 8 : 4) == 0 && ((uintptr_t)d->residual & 15) == 0),
               "strip_linear: residual alignment");
  HFTT_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "strip_linear: drop_p out of range");
  HFTT_REQUIRE(d->res_mod >= 0, "strip_linear: res_mod must be >= 0");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool xbf = d->flags & HFTT_SL_X_BF16, cbf = d->flags & HFTT_SL_C_BF16;
  if (d->ln_gamma != nullptr) {
    HFTT_REQUIRE(xbf && cbf && (d->residual == nullptr || (d->flags & HFTT_SL_RES_BF16)), "strip_linear: the LayerNorm form is all-bf16");
    HFTT_REQUIRE(d->N == 256 && d->ln_beta != nullptr && d->ldc == 256, "strip_linear: LayerNorm needs N == 256 == ldc and beta");
    HFTT_REQUIRE(d->gate == nullptr && !(d->flags & HFTT_SL_RELU), "strip_linear: gate / ReLU cannot be combined with LayerNorm");
    HFTT_REQUIRE(d->pre_ln_out == nullptr || ((uintptr_t)d->pre_ln_out & 15) == 0, "strip_linear: pre_ln_out alignment");
  }
  {                                                   // the persistent software-pipelined form where it applies (strip_gemm2.hip)
    int rc;
    rc = hftt_strip_linear2_try(*d, st);
    if (rc >= 0) return rc;
  }
  if (d->ln_gamma != nullptr) {
    return launch_linear<true, true, true>(*d, st);
  }
  if (cbf) return xbf ? launch_linear<true, true, false>(*d, st) : launch_linear<false, true, false>(*d, st);
  return xbf ? launch_linear<true, false, false>(*d, st) : launch_linear<false, false, false>(*d, st);
}

int check_ffn_x3(const hftt_ffn_desc* d, int mode, const char* what) {
  HFTT_REQUIRE(d->mode == mode, "%s: descriptor mode %d", what, d->mode);
  HFTT_REQUIRE(d->M > 0 && (d->d == 256 || (d->d == 64 && d->p == 128)), "%s: needs d == 256, or d == 64 with p == 128 (got M=%d d=%d p=%d)", what, d->M, d->d, d->p);
  HFTT_REQUIRE((long)d->M * d->p < (1L << 33), "%s: M*p too large for the 32-bit dropout pair index", what);
  HFTT_REQUIRE(d->x != nullptr && d->w != nullptr && d->y != nullptr, "%s: null operand", what);
  HFTT_REQUIRE(d->ldx % 4 == 0 && d->ldy % 4 == 0 && (d->h_out == nullptr || d->ldh % 4 == 0) && (d->residual == nullptr || d->ldr % 4 == 0) && (d->gate == nullptr || d->ldg % 4 == 0),
               "%s: rows must be 16-byte aligned", what);
  HFTT_REQUIRE((((uintptr_t)d->x | (uintptr_t)d->y | (uintptr_t)d->w | (uintptr_t)d->h_out | (uintptr_t)d->gate | (uintptr_t)d->residual | (uintptr_t)d->pre_ln_out) & 15) == 0,
               "%s: operands must be 16-byte aligned", what);
  HFTT_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "%s: drop_p out of range", what);
  return 0;
}

extern "C" int hftt_ffn_res_ln_fwd(const hftt_ffn_desc* d, void* stream) {
  if (d != nullptr && (d->flags & (HFTT_SL_X3_F16 | HFTT_SL_X3_BF16))) {
    if (int rc = check_ffn_x3(d, 0, "ffn_res_ln_fwd")) return rc;
    HFTT_REQUIRE(d->ln_gamma != nullptr && d->ln_beta != nullptr && d->ldy == d->d, "ffn_res_ln_fwd: needs gamma, beta and ldy == d");
    return hftt_x3_strip_mlp(*d, reinterpret_cast<hipStream_t>(stream));
  }
  if (d != nullptr && d->mode == 0) { const int rc = hftt_bs_strip_mlp_try(*d, reinterpret_cast<hipStream_t>(stream)); if (rc >= 0) return rc; }
  if (int rc = check_ffn(d, 0, "ffn_res_ln_fwd")) return rc;
  HFTT_REQUIRE(d->ln_gamma != nullptr && d->ln_beta != nullptr && d->ldy == 256, "ffn_res_ln_fwd: needs gamma, beta and ldy == 256");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rc2 = hftt_strip_mlp2_try(*d, st);
  if (rc2 >= 0) return rc2;
  return launch_mlp<0>(*d, st);
}

extern "C" int hftt_attn_out_ffn_fwd(const hftt_strip_desc* o, const hftt_ffn_desc* f, void* stream) {
  HFTT_REQUIRE(o != nullptr && f != nullptr, "attn_out_ffn_fwd: null descriptor");
  return hftt_x3_attn_out_ffn(*o, *f, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int hftt_ffn_bwd_dx(const hftt_ffn_desc* d, void* stream) {
  if (d != nullptr && (d->flags & (HFTT_SL_X3_F16 | HFTT_SL_X3_BF16))) {
    if (int rc = check_ffn_x3(d, 1, "ffn_bwd_dx")) return rc;
    HFTT_REQUIRE(d->gate != nullptr && d->h_out != nullptr, "ffn_bwd_dx: needs the stored hidden (gate) and the dh output");
    return hftt_x3_strip_mlp(*d, reinterpret_cast<hipStream_t>(stream));
  }
  if (d != nullptr && d->mode == 1) { const int rc = hftt_bs_strip_mlp_try(*d, reinterpret_cast<hipStream_t>(stream)); if (rc >= 0) return rc; }
  if (int rc = check_ffn(d, 1, "ffn_bwd_dx")) return rc;
  HFTT_REQUIRE(d->gate != nullptr && d->ldg % 8 == 0 && ((uintptr_t)d->gate & 15) == 0, "ffn_bwd_dx: needs the stored hidden (bf16, 16-byte aligned rows)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rc2 = hftt_strip_mlp2_try(*d, st);
  if (rc2 >= 0) return rc2;
  return launch_mlp<1>(*d, st);
}
