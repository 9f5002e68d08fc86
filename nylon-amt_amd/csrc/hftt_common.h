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
__device__ __forceinline__ uint32_t hftt_hash(uint64_t seed, uint32_t site, uint64_t idx) {
  // key: one splitmix64 round over (seed, site) -- wave-uniform, so it runs on the scalar ALU once per kernel
  uint64_t k = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1u);
  k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull;
  k ^= k >> 27; k *= 0x94D049BB133111EBull;
  k ^= k >> 31;
  // per element: a 32-bit two-multiply mixer (64-bit multiplies cost ~4x on the vector ALU and the dropout sites hash
  // ~2e9 elements per paper-size step); the high index word only matters beyond 2^32 elements per site
  const uint32_t hi = (uint32_t)(idx >> 32);
  uint32_t x = ((uint32_t)idx + (uint32_t)k) ^ (hi ^ (hi << 16));
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x ^= (uint32_t)(k >> 32); x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
// keep threshold: keep iff hash < thr, thr = (1-p)*2^32 (clamped)
__host__ __device__ inline uint32_t hftt_keep_thr(float p) {
  double k = (1.0 - (double)p) * 4294967296.0;
  if (k >= 4294967295.0) return 0xFFFFFFFFu;
  if (k <= 0.0) return 0u;
  return (uint32_t)k;
}
__device__ __forceinline__ bool hftt_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thr) {
  return hftt_hash(seed, site, idx) < thr;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
