// Split-operand ("x3") arithmetic shared by the kernels of the 1e-3 precision mode (gfx950 / CDNA4 only).
//
// An fp32 operand x is carried as two 16-bit values x ~ hi + lo and a product a.b as THREE matrix-core passes
//     a.b ~ a_lo.b_hi + a_hi.b_lo + a_hi.b_hi            (fp32 accumulate; the lo.lo term is below the representation error)
// on the bf16-rate MFMA pipe (3/16 of the cost of the exact-fp32 MFMA the round-1 parity mode used).  Two element types:
//   * X3_F16  (npass 2): hi, lo are fp16 (RNE).  hi + lo carries ~22 significant bits (2^-22 |x|, or 3e-8 absolute once lo is a
//     subnormal): indistinguishable from fp32 on the reference model (tests/dev_precision_emul.py: outputs within 1.3e-4, the fp32
//     graph itself is 7e-5 from fp64).  Used for every FORWARD product (and the backward's recomputation of the attention scores,
//     which must reproduce the forward's bit for bit).  Range: fp16 saturates at 65504; the conversion clamps instead of producing
//     infinities (activations of this model stay below ~2e3, the attention scores live in fp32 accumulators).
//   * X3_BF16 (npass 4): hi, lo are bf16 (~16 significant bits, fp32's exponent range).  Used for every product that has a GRADIENT
//     operand (values of 1e-4 .. 1e-10 would be subnormal or zero in fp16).  Measured on the forward it is NOT enough for the
//     reference's first encoder layer (logits of ~1e5: 1.2e-3 on the velocity logits), which is why the forward is fp16.
#pragma once
#include "hftt_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int X3_F16 = 2;      // the values double as the descriptors' `npass` codes
constexpr int X3_BF16 = 4;
// X3_BF16H: bf16 halves where the GRADIENT operand of the product enters as its bf16 rounding only (hi half; two MFMA passes
// hi.lo + hi.hi against the other operand's pair).  A gradient only has to point the right way: its 2^-9 zero-mean rounding moves no
// gradient tensor's cosine against the exact-fp32 mode below 0.9999 (tests/test_paper_bf16_gpu.py), the saved forward operand keeps 16 bits.
constexpr int X3_BF16H = 5;

template <int E>
struct X3;

template <>
struct X3<X3_F16> {
  // two fp32 -> packed fp16 pair (v_cvt_pk_f16_f32, round to nearest even), saturating
  static __device__ __forceinline__ unsigned pk(float a, float b) {
    const f32x2_t v = {__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  }
  // plain conversion (no saturation): the lo halves (below one fp16 ulp of hi once the VALUE was clamped) and values known to be in range
  static __device__ __forceinline__ unsigned pk_lo(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  }
  static __device__ __forceinline__ void unpk(unsigned p, float& a, float& b) {
    const f32x2_t v = __builtin_convertvector(__builtin_bit_cast(f16x2_t, p), f32x2_t);
    a = v[0]; b = v[1];
  }
  // the packed lo halves of (a, b) whose packed hi halves are `hi`: a - hi in ONE v_fma_mix_f32 per element (the fp16 half is read in place
  // as an FMA operand; the difference is exact either way) instead of a conversion back to fp32 plus a subtraction -- the split runs inside
  // every VALU-bound loop of the x3 kernels (per pair: cvt_pk, 2 x fma_mix, cvt_pk)
  static __device__ __forceinline__ unsigned lo_of(float a, float b, unsigned hi) {
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
    return pk_lo(la, lb);
  }
  static __device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

template <>
struct X3<X3_BF16> {
  static __device__ __forceinline__ unsigned pk(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  }
  static __device__ __forceinline__ unsigned pk_lo(float a, float b) { return pk(a, b); }
  static __device__ __forceinline__ void unpk(unsigned p, float& a, float& b) {
    a = __uint_as_float(p << 16); b = __uint_as_float(p & 0xFFFF0000u);
  }
  static __device__ __forceinline__ unsigned lo_of(float a, float b, unsigned hi) {
    float ha, hb;
    unpk(hi, ha, hb);
    return pk(a - ha, b - hb);
  }
  static __device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) { return mfma32(a, b, c); }
  static __device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) { return mfma16(a, b, c); }
};

template <>
struct X3<X3_BF16H> : X3<X3_BF16> {};

// (a, b) -> packed hi pair + packed lo pair
// The VALUE is saturated once (fp16: +-65504; bf16: nothing to do), then split: a finite input always gives a finite pair -- no infinities
// are manufactured inside a GEMM whatever a diverging run feeds it (per pair: 2 x v_med3, cvt_pk, 2 x fma_mix, cvt_pk).  v_med3_f32 returns a
// finite value for a NaN input too (the minimum of the other two operands), so THIS form does not carry a NaN: on the activation stream the
// fp32 residual / LayerNorm path next to every GEMM does (tensors stay fp32 between kernels); the per-step weight preparation, whose
// operands have no such path, uses x3_split2_checked below.
template <int E>
__device__ __forceinline__ void x3_split2(float a, float b, unsigned& hi, unsigned& lo) {
  if (E == X3_F16) {
    a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
    b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
  }
  hi = X3<E>::pk_lo(a, b);
  lo = X3<E>::lo_of(a, b, hi);
}
// the same for the weight-preparation kernels (once per step over the parameters: not a hot path): a NaN or infinite parameter comes out as
// a NaN lo half, so a diverged optimizer state poisons every product of the next forward instead of being laundered into +-65504
template <int E>
__device__ __forceinline__ void x3_split2_checked(float a, float b, unsigned& hi, unsigned& lo) {
  x3_split2<E>(a, b, hi, lo);
  const unsigned nan_lo = (E == X3_F16) ? 0x7E00u : 0x7FC0u;
  if (!(a - a == 0.f)) lo = (lo & 0xFFFF0000u) | nan_lo;
  if (!(b - b == 0.f)) lo = (lo & 0x0000FFFFu) | (nan_lo << 16);
}
// the same for values known to be finite and inside the element type's range (softmax probabilities): no saturation step
template <int E>
__device__ __forceinline__ void x3_split2_nc(float a, float b, unsigned& hi, unsigned& lo) {
  hi = X3<E>::pk_lo(a, b);
  lo = X3<E>::lo_of(a, b, hi);
}
template <int E>
__device__ __forceinline__ void x3_split8_nc(const float* v, bf16x8& hi, bf16x8& lo) {
  uint4 h, l;
  x3_split2_nc<E>(v[0], v[1], h.x, l.x); x3_split2_nc<E>(v[2], v[3], h.y, l.y);
  x3_split2_nc<E>(v[4], v[5], h.z, l.z); x3_split2_nc<E>(v[6], v[7], h.w, l.w);
  hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}
// four consecutive elements -> 8 bytes of the hi plane + 8 bytes of the lo plane
template <int E>
__device__ __forceinline__ void x3_split4(const float4& f, uint2& hi, uint2& lo) {
  x3_split2<E>(f.x, f.y, hi.x, lo.x);
  x3_split2<E>(f.z, f.w, hi.y, lo.y);
}
// eight consecutive elements -> one MFMA fragment register quad of each plane
template <int E>
__device__ __forceinline__ void x3_split8(const float* v, bf16x8& hi, bf16x8& lo) {
  uint4 h, l;
  x3_split2<E>(v[0], v[1], h.x, l.x); x3_split2<E>(v[2], v[3], h.y, l.y);
  x3_split2<E>(v[4], v[5], h.z, l.z); x3_split2<E>(v[6], v[7], h.w, l.w);
  hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}

template <int E>
__device__ __forceinline__ void x3_split8_checked(const float* v, bf16x8& hi, bf16x8& lo) {
  uint4 h, l;
  x3_split2_checked<E>(v[0], v[1], h.x, l.x); x3_split2_checked<E>(v[2], v[3], h.y, l.y);
  x3_split2_checked<E>(v[4], v[5], h.z, l.z); x3_split2_checked<E>(v[6], v[7], h.w, l.w);
  hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}

// D = A.B + C in three passes, small terms first
template <int E>
__device__ __forceinline__ f32x16 x3_mma(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
  c = X3<E>::mma(al, bh, c);
  c = X3<E>::mma(ah, bl, c);
  return X3<E>::mma(ah, bh, c);
}
template <int E>
__device__ __forceinline__ f32x4 x3_mma16(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
  c = X3<E>::mma16(al, bh, c);
  c = X3<E>::mma16(ah, bl, c);
  return X3<E>::mma16(ah, bh, c);
}
