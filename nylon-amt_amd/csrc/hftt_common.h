// Device-side helpers shared by all hFT-Transformer kernels (gfx950 / CDNA4 only).
// Lane maps used here were verified on MI355X by tools/probe_mfma.hip (T1..T5 all PASS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HFTT_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// fp32 -> bf16 (round to nearest even; lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short h) {
  return __builtin_bit_cast(float, ((unsigned int)h) << 16);
}
// split x = hi + lo (+ O(2^-17 |x|)), both bf16: the "bf16x3" operand pair
__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// D = A*B + C with split operands: hi*hi + hi*lo + lo*hi (NPASS==3) or hi*hi only (NPASS==1)
template <int NPASS>
__device__ __forceinline__ f32x16 mfma32_split(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
  if (NPASS == 3) {
    c = mfma32(al, bh, c);
    c = mfma32(ah, bl, c);
  }
  return mfma32(ah, bh, c);
}
template <int NPASS>
__device__ __forceinline__ f32x4 mfma16_split(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
  if (NPASS == 3) {
    c = mfma16(al, bh, c);
    c = mfma16(ah, bl, c);
  }
  return mfma16(ah, bh, c);
}

// exact-fp32 MFMA (parity mode): D = A*B + C with ONE f32 per lane per operand,
//   32x32x2: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31];  16x16x4: A[i = lane&15][k = lane>>4], B[k][j = lane&15]
// (bit-for-bit a k-ordered fmaf chain; runs at the fp32 vector rate = 1/16 of the bf16 MFMA rate)
__device__ __forceinline__ f32x16 mfma32_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16_f32(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// row index inside a 32x32 accumulator tile for register g of lane-half h (col = lane&31)
__device__ __forceinline__ int acc_row32(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// LDS reads
__device__ __forceinline__ bf16x8 lds_read_b128(const unsigned short* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
// transposed 4x16 block read (ds_read_b64_tr_b16): p = this lane's address (row q=(lane&15)>>2, cols 4*(lane&3)..)
__device__ __forceinline__ bf16x4 lds_read_tr16(const unsigned short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(HFTT_LDS_PTR(bf16x4, p));
}
__device__ __forceinline__ bf16x8 join4(bf16x4 a, bf16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// ---- tensors stored as fp32 or as bf16 (io_flags): 4 consecutive elements at ELEMENT offset `off` ----
__device__ __forceinline__ float4 hftt_load4(const float* base, bool bf, long off) {
  if (bf) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
    return make_float4(bf2f(u.x & 0xFFFFu), bf2f(u.x >> 16), bf2f(u.y & 0xFFFFu), bf2f(u.y >> 16));
  }
  return *reinterpret_cast<const float4*>(base + off);
}
__device__ __forceinline__ void hftt_store4(float* base, bool bf, long off, float a, float b, float c, float d) {
  if (bf) {
    uint2 u;
    u.x = f2bf(a) | ((unsigned)f2bf(b) << 16); u.y = f2bf(c) | ((unsigned)f2bf(d) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = u;
  } else {
    *reinterpret_cast<float4*>(base + off) = make_float4(a, b, c, d);
  }
}
__device__ __forceinline__ void hftt_store1(float* base, bool bf, long off, float a) {
  if (bf) reinterpret_cast<unsigned short*>(base)[off] = f2bf(a);
  else base[off] = a;
}

// ---- counter-based dropout RNG: keep(seed, site, element index) -- identical in forward and backward ----
// key: one splitmix64 round over (seed, site) -- wave-uniform, so it runs on the scalar ALU once per kernel
__device__ __forceinline__ uint64_t hftt_hash_key(uint64_t seed, uint32_t site) {
  uint64_t k = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1u);
  k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull;
  k ^= k >> 27; k *= 0x94D049BB133111EBull;
  k ^= k >> 31;
  return k;
}
// the high index word's contribution (it only matters beyond 2^32 hashed indices per site): hoistable when a loop walks the low word
__device__ __forceinline__ uint32_t hftt_hash_himix(uint32_t hi) { return hi ^ (hi << 16); }
// per index: a 32-bit two-multiply mixer (64-bit multiplies cost ~4x on the vector ALU and the dropout sites hash ~2e9 elements
// per paper-size step; even v_mul_lo_u32 is a quarter-rate instruction, which is why kernels share one hash between two elements)
__device__ __forceinline__ uint32_t hftt_hash_mix(uint64_t k, uint32_t lo, uint32_t himix) {
  uint32_t x = (lo + (uint32_t)k) ^ himix;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x ^= (uint32_t)(k >> 32); x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t hftt_hash(uint64_t seed, uint32_t site, uint64_t idx) {
  return hftt_hash_mix(hftt_hash_key(seed, site), (uint32_t)idx, hftt_hash_himix((uint32_t)(idx >> 32)));
}
// One 32-bit hash serves FOUR consecutive elements (4q .. 4q+3): each takes one BYTE of the word, compared with an 8-bit threshold
//   keep(idx) = byte (idx & 3) of hash(idx >> 2) < thr,   thr = round((1 - p) * 256)
// so the keep probability is quantised to thr / 256 (p = 0.1 -> thr 230, drop rate 0.1016).  Kept elements are scaled by 256 / thr, the
// reciprocal of the probability actually applied (hftt_keep_scale), so the expected value of a dropped tensor is exactly the input --
// nn.Dropout's contract (the reference's 1 / (1 - p) is the same number whenever (1 - p) * 256 is an integer, e.g. p = 0.25, 0.5).
// v_mul_lo_u32 issues at quarter rate, so the two multiplies of the mixer
// are most of a hash: kernels whose lanes own adjacent elements hash once per quad; every other site calls hftt_keep per element
// and gets the same decisions.
__host__ __device__ inline uint32_t hftt_keep_thr(float p) {
  double k = (1.0 - (double)p) * 256.0 + 0.5;
  if (k >= 256.0) return 256u;
  if (k <= 0.0) return 0u;
  return (uint32_t)k;
}
// scale of the kept elements: 1 / (keep probability actually applied) = 256 / thr   (1 when nothing is dropped)
__host__ __device__ inline float hftt_keep_scale(float p) {
  if (!(p > 0.f)) return 1.0f;
  const uint32_t thr = hftt_keep_thr(p);
  return thr > 0u ? 256.0f / (float)thr : 0.0f;
}
// energy = (Q K^T) / sqrt(dh) applied to one raw product, the SAME way in a forward and the backward that recomputes its P:
// dh = 64: x / 8 exactly;  dh = 32: the fp32 mode divides (the reference's own operation, model_spec2midi.py:354), the bf16 mode multiplies.
template <int DH, bool EXACT>
__device__ __forceinline__ float hftt_attn_scaled(float x) {
  if constexpr (DH == 64) return x * 0.125f;
  else if constexpr (EXACT) return x / sqrtf((float)DH);
  else return x * (1.0f / sqrtf((float)DH));
}
__device__ __forceinline__ bool hftt_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thr) {
  const uint32_t w = hftt_hash(seed, site, idx >> 2);
  return ((w >> (8u * ((uint32_t)idx & 3u))) & 0xFFu) < thr;
}
// the four decisions of the quad (4q .. 4q+3): bit e = keep(4q + e)
__device__ __forceinline__ uint32_t hftt_keep_quad(uint64_t seed, uint32_t site, uint64_t q, uint32_t thr) {
  const uint32_t w = hftt_hash(seed, site, q);
  return ((w & 0xFFu) < thr ? 1u : 0u) | (((w >> 8) & 0xFFu) < thr ? 2u : 0u) | (((w >> 16) & 0xFFu) < thr ? 4u : 0u) | ((w >> 24) < thr ? 8u : 0u);
}

// ---- cross-lane traffic without the LDS crossbar ----------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS instruction + an lgkmcnt wait per use); DPP / permlane / readlane stay on the VALU.
template <int CTRL>
__device__ __forceinline__ float hftt_dpp(float v) {      // CTRL: 0xB1 quad_perm[1,0,3,2]  0x4E quad_perm[2,3,0,1]  0x141 row_half_mirror  0x140 row_mirror
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float hftt_readlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum / max over the 64 lanes, result wave-uniform (four DPP steps inside each 16-lane row, then four readlanes)
__device__ __forceinline__ float wave_sum(float v) {
  v += hftt_dpp<0xB1>(v); v += hftt_dpp<0x4E>(v); v += hftt_dpp<0x141>(v); v += hftt_dpp<0x140>(v);
  return (hftt_readlane(v, 0) + hftt_readlane(v, 16)) + (hftt_readlane(v, 32) + hftt_readlane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, hftt_dpp<0xB1>(v)); v = fmaxf(v, hftt_dpp<0x4E>(v)); v = fmaxf(v, hftt_dpp<0x141>(v)); v = fmaxf(v, hftt_dpp<0x140>(v));
  return fmaxf(fmaxf(hftt_readlane(v, 0), hftt_readlane(v, 16)), fmaxf(hftt_readlane(v, 32), hftt_readlane(v, 48)));
}
// sum over aligned groups of G lanes (G = 2, 4, 8 or 16), result in every lane of the group
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group_sum: G must be a power of two <= 16");
  if (G >= 2) v += hftt_dpp<0xB1>(v);
  if (G >= 4) v += hftt_dpp<0x4E>(v);
  if (G >= 8) v += hftt_dpp<0x141>(v);
  if (G >= 16) v += hftt_dpp<0x140>(v);
  return v;
}
// lane l <-> lane l ^ 32 (v_permlane32_swap, gfx950): both halves end up with op(own, partner).  Inline asm: hipcc 7.2 folds the
// two results of __builtin_amdgcn_permlane32_swap into one value.  After the swap a = [v.lo | v.lo], b = [v.hi | v.hi].
// (s_nop: a VALU write of an operand needs wait states before / after the swap.)
__device__ __forceinline__ void hftt_swap32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xor32_sum(float v) {
  float a = v, b = v;
  hftt_swap32(a, b);
  return a + b;
}
__device__ __forceinline__ float xor32_max(float v) {
  float a = v, b = v;
  hftt_swap32(a, b);
  return fmaxf(a, b);
}
// Accumulator layouts put adjacent COLUMNS in lanes 2i / 2i+1 and two ROWS (a, b) in one lane.  pair_rows_to_cols() turns that
// into one packed bf16 pair of adjacent columns per lane: the even lane gets row a = [own a | partner a], the odd lane row b =
// [partner b | own b].  One cvt_pk + one DPP move + one byte permute for two elements.
// the same from an already packed pair pk = [a | b] (lo16 = row a, hi16 = row b of this lane's column)
__device__ __forceinline__ unsigned packed_rows_to_cols(unsigned pk, bool odd) {
  const unsigned q = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0xB1, 0xF, 0xF, true);   // partner's pair
  return __builtin_amdgcn_perm(pk, q, odd ? 0x07060302u : 0x01000504u);
}
__device__ __forceinline__ unsigned pair_rows_to_cols(float a, float b, bool odd) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {a, b};
  const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2_t));      // lo16 = a, hi16 = b
  const unsigned q = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0xB1, 0xF, 0xF, true);   // partner's pair
  // v_perm_b32: selector bytes 0-3 pick from the second operand (q), 4-7 from the first (pk)
  return __builtin_amdgcn_perm(pk, q, odd ? 0x07060302u : 0x01000504u);
}
