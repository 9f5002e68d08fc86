// Helpers shared by the pipelined strip kernels (strip_gemm2.hip, bs_strip.hip): ring constants, static_for, LDS-DMA and store asm,
// bf16 packing, the 16-element dropout, the slot's MFMA loops, the LayerNorm row pass.  Included inside an anonymous namespace.
#pragma once
constexpr int SLOT_BYTES = 16384;
constexpr int NSLOT = 4;
constexpr int RING_BYTES = NSLOT * SLOT_BYTES;
constexpr int FILL_AHEAD = NSLOT - 1;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a compile-time constant in every iteration.
// (#pragma unroll on the slot loops was only honoured by a factor of four here -- the prefetch register arrays then had a run-time
// index and went to scratch memory.)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// 16 bytes in four VGPRs as a native vector (HIP's uint4 is a struct: inline-asm operands of struct type go through memory)
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

// four LDS-DMA instructions (16 B per lane each): sources gsrc + {0, 1, 2, 3} KiB, LDS destinations lds_dst + {0, 1, 2, 3} KiB (+ 16 * lane).
// The instruction offset advances BOTH addresses (LDS address = M0 + offset + 16 * lane; measured: adding it to M0 as well put the
// fragments 1 KiB too far).  M0 (compiler-reserved) is saved once and restored once in the same statement.
__device__ __forceinline__ void glds16x4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst));
}
// prefetch load: an ordinary 16-byte load (exactly one global_load_dwordx4), counted by the caller
__device__ __forceinline__ void pload16(u4v& dst, const void* p) { dst = *reinterpret_cast<const u4v*>(p); }
// store from asm (no destination registers: nothing the allocator could get wrong), counted by the caller
__device__ __forceinline__ void astore16(void* p, const u4v& v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v)); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// s_waitcnt vmcnt(k) for a wave-uniform run-time n, with k = the largest value of a 16-entry grid that is <= n (waiting for a few
// more operations than necessary is always safe).  The immediate must be static, so this is a depth-4 branch tree; the plain
// 64-way switch it replaces was lowered to a chain of 64 compare-and-branch pairs, ~1,000 cycles per slot.
#define HFTT_WAITVM(k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  if (n < 16) {
    if (n < 8) {
      if (n < 4) { if (n < 2) HFTT_WAITVM(0); else HFTT_WAITVM(2); }
      else { if (n < 6) HFTT_WAITVM(4); else HFTT_WAITVM(6); }
    } else {
      if (n < 12) { if (n < 10) HFTT_WAITVM(8); else HFTT_WAITVM(10); }
      else { if (n < 14) HFTT_WAITVM(12); else HFTT_WAITVM(14); }
    }
  } else {
    if (n < 32) {
      if (n < 24) { if (n < 20) HFTT_WAITVM(16); else HFTT_WAITVM(20); }
      else { if (n < 28) HFTT_WAITVM(24); else HFTT_WAITVM(28); }
    } else {
      if (n < 48) { if (n < 40) HFTT_WAITVM(32); else HFTT_WAITVM(40); }
      else { if (n < 63) HFTT_WAITVM(48); else HFTT_WAITVM(63); }
    }
  }
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2_t));
}
__device__ __forceinline__ u4v pack8u(const float* v) {
  u4v o;
  o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
  return o;
}
__device__ __forceinline__ void unpack8(u4v q, float* v) {
  v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xFFFF0000u);
  v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xFFFF0000u);
  v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xFFFF0000u);
  v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xFFFF0000u);
}
__device__ __forceinline__ void lds16f(const float* p, float* v) {
#pragma unroll
  for (int q = 0; q < 4; q++) { const float4 t = reinterpret_cast<const float4*>(p)[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}
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
// the same for 8 consecutive elements (two hash quads): the backward of a dropout applied to a strip chunk as it is loaded
__device__ __forceinline__ void drop8(float* v, uint64_t seed, uint32_t site, uint64_t q0, uint32_t thr, float inv_keep) {
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const uint32_t w = hftt_hash(seed, site, q0 + e);
    uint32_t m[4];
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m[0]) : "v"((w & 0xFFu) - thr));
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m[1]) : "v"(((w >> 8) & 0xFFu) - thr));
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m[2]) : "v"(((w >> 16) & 0xFFu) - thr));
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m[3]) : "v"((w >> 24) - thr));
#pragma unroll
    for (int f = 0; f < 4; f++) v[4 * e + f] = __uint_as_float(__float_as_uint(v[4 * e + f] * inv_keep) & m[f]);
  }
}
__device__ __forceinline__ bf16x8 as_frag(const u4v& u) { return __builtin_bit_cast(bf16x8, u); }

// The 16 weight fragments of a slot -> 16 MFMAs.  With ONE wave per SIMD nobody else covers an LDS round trip, and left alone hipcc
// keeps only one or two ds_read_b128 in flight (~120 cycles per MFMA instead of 32).  The group barriers pin the shape: six reads
// up front, then one read behind every MFMA, so each fragment is requested ~6 MFMAs (~190 cycles) before it is consumed.
template <int AHEAD = 6, typename F>
__device__ __forceinline__ void slot_mfmas(const unsigned char* slot, F&& mfma_i, bool noread = false) {
  bf16x8 fr[16];
#if defined(HFTT_MLP2_CT) && (HFTT_MLP2_CT & 8)
  noread = true;
#endif
  if (noread) {                                       // timing experiment (HFTT_STRIP2_DEBUG bit 1024): one fragment read feeds all sixteen MFMAs
    const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(slot);
#pragma unroll
    for (int i = 0; i < 16; i++) mfma_i(i, f0);
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; i++) fr[i] = *reinterpret_cast<const bf16x8*>(slot + i * 1024);
#pragma unroll
  for (int i = 0; i < 16; i++) mfma_i(i, fr[i]);
  __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
  for (int k = 0; k < 16 - AHEAD; k++) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
}

// The same, with the slot's memory work (`side(i)`, called behind MFMA i) in program order BETWEEN the MFMAs: a wave alone on its SIMD
// issues in order, so whatever follows the last MFMA is paid in full, while an instruction placed between two MFMAs issues in the 24 of
// 32 cycles the matrix pipe leaves free.
template <typename F, typename G>
__device__ __forceinline__ void slot_mfmas_mix(const unsigned char* slot, F&& mfma_i, G&& side) {
  bf16x8 fr[16];
#pragma unroll
  for (int i = 0; i < 6; i++) fr[i] = *reinterpret_cast<const bf16x8*>(slot + i * 1024);
  static_for<16>([&](auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    mfma_i(i, fr[i]);
    if (i + 6 < 16) fr[i + 6] = *reinterpret_cast<const bf16x8*>(slot + (i + 6) * 1024);
    side(i_c);
    __builtin_amdgcn_sched_barrier(0);               // keep this MFMA, its refill read and its share of the side work together
  });
}

// Whole-line stores of bf16 results (round 5).  Stored straight from the accumulator layout one instruction scatters 64 16-byte pieces over
// 64 rows -- eight instructions per 128-byte line -- and the CU's store path then moves ~7 B/clk (MI355X_MICROARCH.md, store-issue bound;
// tools/ablate_mlp2.sh: the training form's final epilogue is 100 of 349 us against 32 for the inference form, which stores a third of the
// bytes).  A PAIR of adjacent 32-column tiles of a wave's 32 tokens is 32 rows x 128 bytes: the lanes put their halves into a wave-private
// LDS patch (row pitch 144 B: the 16-byte pieces of 8 rows fall on 8 different bank quads) and read it back 8 lanes per row, so each of the
// four store instructions of the pair writes 8 complete lines.  LDS operations of one wave execute in order: no barrier.
constexpr int PATCH_PITCH = 144;
constexpr int PATCH_BYTES = 32 * PATCH_PITCH;          // per wave and patch
// lane (j, h): token j, its 16 features (two 16-byte halves a, b) of tile `odd` (0 / 1) of the pair
__device__ __forceinline__ void patch_put(unsigned char* patch, int j, int h, int odd, const u4v& a, const u4v& b) {
  u4v* w = reinterpret_cast<u4v*>(patch + j * PATCH_PITCH + odd * 64 + h * 32);
  w[0] = a; w[1] = b;
}
// piece k = 0..3 of the pair: row (lane >> 3) + 8 k, 16-byte chunk lane & 7 of the row's 128 bytes
__device__ __forceinline__ u4v patch_get(const unsigned char* patch, int lane, int k) {
  return *reinterpret_cast<const u4v*>(patch + ((lane >> 3) + 8 * k) * PATCH_PITCH + (lane & 7) * 16);
}
// byte offset of that piece from the wave's first row: ld in elements (bf16), pr = pair number (64 columns each)
__device__ __forceinline__ unsigned patch_off(int lane, int k, int ld, int pr) {
  return (unsigned)(((lane >> 3) + 8 * k) * ld * 2 + pr * 128 + (lane & 7) * 16);
}
// store with a wave-uniform base and a 32-bit lane offset (no 64-bit vector address arithmetic per piece)
__device__ __forceinline__ void astore16s(const void* base, unsigned off, const u4v& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base));
}
// Whole-line LOADS, the mirror image: a wave's 32 rows x 128 bytes (a pair of tiles) arrive as four line pieces (piece k: row (lane >> 3) + 8 k,
// chunk lane & 7 -- each load instruction reads 8 complete lines instead of touching 64 rows), go into the patch as they are and come back as
// the lane's own two 16-byte halves of each tile (the MFMA B operand layout).
__device__ __forceinline__ u4v aload16s(const void* base, unsigned off) {
  return *reinterpret_cast<const u4v*>(reinterpret_cast<const unsigned char*>(base) + off);
}
__device__ __forceinline__ void patch_put_lines(unsigned char* patch, int lane, const u4v& l0, const u4v& l1, const u4v& l2, const u4v& l3) {
  unsigned char* w = patch + (lane >> 3) * PATCH_PITCH + (lane & 7) * 16;
  *reinterpret_cast<u4v*>(w) = l0;
  *reinterpret_cast<u4v*>(w + 8 * PATCH_PITCH) = l1;
  *reinterpret_cast<u4v*>(w + 16 * PATCH_PITCH) = l2;
  *reinterpret_cast<u4v*>(w + 24 * PATCH_PITCH) = l3;
}
// lane (j, h): halves (tile A: a0, a1; tile B: b0, b1) of its token's row
__device__ __forceinline__ void patch_get_rows(const unsigned char* patch, int j, int h, u4v& a0, u4v& a1, u4v& b0, u4v& b1) {
  const u4v* r = reinterpret_cast<const u4v*>(patch + j * PATCH_PITCH + h * 32);
  a0 = r[0]; a1 = r[1]; b0 = r[4]; b1 = r[5];
}
// the four pieces of a finished pair at once
__device__ __forceinline__ void patch_flush(const unsigned char* patch, int lane, const void* wave_base, int ld, int pr) {
#pragma unroll
  for (int k = 0; k < 4; k++) astore16s(wave_base, patch_off(lane, k, ld, pr), patch_get(patch, lane, k));
}

// element offset (inside the lane's 16-feature-per-tile row view) of load / store piece i = 0..15: tile (i >> 1), 8-element half (i & 1)
__device__ __forceinline__ long piece_off(int i) { return (long)(i >> 1) * 32 + (i & 1) * 8; }

// LayerNorm over the 256 features of the lane's token (lane: 128 of them in acc, partner lane ^ 32 the rest); the normalised row goes to
// emit(tile, first half, second half) (deferred or immediate stores), the pre-LayerNorm sum (training) is stored at once
template <typename F, typename E, typename P>
__device__ __forceinline__ void ln_rows(f32x16 (&acc)[8], const float* gamma_lds, const float* beta_lds, int h,
                                        float* mean_out, float* rstd_out, long tok, bool wave_ok, bool has_pre, unsigned short* pre_row, F&& count_issue, E&& emit,
                                        P&& emit_pre) {
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
  if (wave_ok && h == 0) {
    if (mean_out != nullptr) mean_out[tok] = mean;
    if (rstd_out != nullptr) rstd_out[tok] = rstd;
  }
#pragma unroll
  for (int ot = 0; ot < 8; ot++) {
    float v[16], ga[16], be[16];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = acc[ot][q];
    if (has_pre && wave_ok) {                       // (both wave-uniform: the issue count stays a scalar)
      const u4v a = pack8u(v), b = pack8u(v + 8);
      if (!emit_pre(ot, a, b)) {                      // (false: the caller has no patch -- row pieces from here)
        astore16(pre_row + piece_off(2 * ot), a);
        astore16(pre_row + piece_off(2 * ot + 1), b);
      }
      count_issue(2);
    }
    lds16f(gamma_lds + ot * 32 + 16 * h, ga);
    lds16f(beta_lds + ot * 32 + 16 * h, be);
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = (v[q] - mean) * rstd * ga[q] + be[q];
    emit(ot, pack8u(v), pack8u(v + 8));
    __builtin_amdgcn_sched_barrier(0);
  }
}
// (callers without a patch: the pre-LayerNorm rows leave as row pieces from inside)
template <typename F, typename E>
__device__ __forceinline__ void ln_rows(f32x16 (&acc)[8], const float* gamma_lds, const float* beta_lds, int h,
                                        float* mean_out, float* rstd_out, long tok, bool wave_ok, bool has_pre, unsigned short* pre_row, F&& count_issue, E&& emit) {
  ln_rows(acc, gamma_lds, beta_lds, h, mean_out, rstd_out, tok, wave_ok, has_pre, pre_row, static_cast<F&&>(count_issue), static_cast<E&&>(emit),
          [](int, const u4v&, const u4v&) { return false; });
}

