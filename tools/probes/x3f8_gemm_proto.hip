// GPU box prototype (not product): one x3 product  C = W . x^T  with the hi.hi pass on the fp16 MFMA and the two cross terms on the block-scaled
// fp8 MFMA, conversions done in the kernel the way a strip kernel would (lane = one row of its operand, 32 of every 64 k-elements), checked against
// float64.  One wave per 32 x 32 tile, operands read straight from global memory: this measures the ARITHMETIC, not speed.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/x3f8_gemm_proto.hip -o /tmp/x3f8 && /tmp/x3f8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// the lane's 32 values of a 64-k group -> fp16 hi (4 x f16x8), e4m3 of hi and of lo (8 words each) with this lane's scale byte for each
struct Ops { f16x8 h[4]; i32x8 h8, l8; int sh, sl; };
__device__ __forceinline__ int scale_exp(float amax) {        // E8M0 byte of 2^(floor(log2 amax) - 8): the block's largest value lands in [256, 512) -> e4m3's top binade is [256, 448]
  if (!(amax > 0.f)) return 127 - 8;
  int e; frexpf(amax, &e);                                     // amax = m 2^e, m in [0.5, 1): floor(log2) = e - 1
  int b = (e - 1) - 7 + 127;                                   // values scaled into [128, 256): one binade of headroom under 448
  return b < 1 ? 1 : (b > 254 ? 254 : b);
}
__device__ __forceinline__ void convert(const float* v, int lane, Ops& o) {
  float hi[32], lo[32];
  for (int i = 0; i < 32; i++) { const _Float16 t = (_Float16)v[i]; hi[i] = (float)t; lo[i] = v[i] - hi[i]; }
  for (int q = 0; q < 4; q++) for (int e = 0; e < 8; e++) o.h[q][e] = (_Float16)hi[8 * q + e];
  // block maxima: bytes 0..15 of this lane and of its partner lane ^ 32 are one scale block, bytes 16..31 the other
  float mh[2] = {0.f, 0.f}, ml[2] = {0.f, 0.f};
  for (int i = 0; i < 32; i++) { mh[i >> 4] = fmaxf(mh[i >> 4], fabsf(hi[i])); ml[i >> 4] = fmaxf(ml[i >> 4], fabsf(lo[i])); }
  for (int b = 0; b < 2; b++) { mh[b] = fmaxf(mh[b], __shfl_xor(mh[b], 32)); ml[b] = fmaxf(ml[b], __shfl_xor(ml[b], 32)); }
  const int eh[2] = {scale_exp(mh[0]), scale_exp(mh[1])}, el[2] = {scale_exp(ml[0]), scale_exp(ml[1])};
  const int half = lane >> 5;
  o.sh = eh[half]; o.sl = el[half];                            // the scale byte the hardware reads from THIS lane: block `half`
  for (int w = 0; w < 8; w++) {
    const int b = w >> 2;                                      // words 0..3 = bytes 0..15 = block 0
    const float sh = __builtin_ldexpf(1.0f, 127 - eh[b]), sl = __builtin_ldexpf(1.0f, 127 - el[b]);
    int ph = 0, pl = 0;
    ph = __builtin_amdgcn_cvt_pk_fp8_f32(hi[4 * w] * sh, hi[4 * w + 1] * sh, ph, false);
    ph = __builtin_amdgcn_cvt_pk_fp8_f32(hi[4 * w + 2] * sh, hi[4 * w + 3] * sh, ph, true);
    pl = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4 * w] * sl, lo[4 * w + 1] * sl, pl, false);
    pl = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4 * w + 2] * sl, lo[4 * w + 3] * sl, pl, true);
    o.h8[w] = ph; o.l8[w] = pl;
  }
}
// MODE 0: three fp16 passes (today's x3)   1: fp16 hi.hi + fp8 cross terms   2: hi.hi only
template <int MODE>
__global__ void k_gemm(const float* __restrict__ x, const float* __restrict__ W, float* __restrict__ C, int M, int N, int K) {
  const int lane = threadIdx.x, r = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  for (int g = 0; g < K / 64; g++) {
    float wv[32], xv[32];
    for (int i = 0; i < 32; i++) {                             // the instruction's own k labelling: k = 32 (i / 16) + 16 half + i % 16
      const int k = 64 * g + 32 * (i >> 4) + 16 * half + (i & 15);
      wv[i] = W[(long)(n0 + r) * K + k]; xv[i] = x[(long)(t0 + r) * K + k];
    }
    Ops w, a;
    convert(wv, lane, w); convert(xv, lane, a);
    for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[q], a.h[q], acc, 0, 0, 0);
    if (MODE == 0) {
      f16x8 wl[4], al[4];
      for (int q = 0; q < 4; q++) for (int e = 0; e < 8; e++) { wl[q][e] = (_Float16)(wv[8 * q + e] - (float)w.h[q][e]); al[q][e] = (_Float16)(xv[8 * q + e] - (float)a.h[q][e]); }
      for (int q = 0; q < 4; q++) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[q], al[q], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[q], a.h[q], acc, 0, 0, 0); }
    } else if (MODE == 1) {
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w.h8, a.l8, acc, 0, 0, 0, w.sh, 0, a.sl);
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w.l8, a.h8, acc, 0, 0, 0, w.sl, 0, a.sh);
    }
  }
  for (int i = 0; i < 16; i++) {                               // rows = outputs, columns = tokens
    const int row = 8 * (i >> 2) + 4 * half + (i & 3);
    C[(long)(t0 + r) * N + n0 + row] = acc[i];
  }
}
int main() {
  const int M = 256, K = 256, N = 512;
  srand(3);
  auto rnd = []() { float s = 0; for (int i = 0; i < 12; i++) s += (float)rand() / RAND_MAX; return s - 6.f; };
  std::vector<float> x((size_t)M * K), W((size_t)N * K), C((size_t)M * N);
  for (int t = 0; t < M; t++) for (int k = 0; k < K; k++) x[(size_t)t * K + k] = rnd() * (k < K / 8 ? 30.f : 1.f);
  for (auto& v : W) v = rnd() * 0.06f;
  std::vector<double> ref((size_t)M * N);
  double den = 0;
  for (int t = 0; t < M; t++) for (int n = 0; n < N; n++) { double s = 0; for (int k = 0; k < K; k++) s += (double)x[(size_t)t * K + k] * (double)W[(size_t)n * K + k]; ref[(size_t)t * N + n] = s; den = fmax(den, fabs(s)); }
  float *dx, *dW, *dC;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  const char* names[3] = {"three fp16 passes", "fp16 hi.hi + fp8 cross terms", "hi.hi only"};
  for (int mode = 0; mode < 3; mode++) {
    hipMemset(dC, 0, C.size() * 4);
    if (mode == 0) hipLaunchKernelGGL(k_gemm<0>, dim3(N / 32, M / 32), dim3(64), 0, 0, dx, dW, dC, M, N, K);
    if (mode == 1) hipLaunchKernelGGL(k_gemm<1>, dim3(N / 32, M / 32), dim3(64), 0, 0, dx, dW, dC, M, N, K);
    if (mode == 2) hipLaunchKernelGGL(k_gemm<2>, dim3(N / 32, M / 32), dim3(64), 0, 0, dx, dW, dC, M, N, K);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (size_t i = 0; i < C.size(); i++) worst = fmax(worst, fabs((double)C[i] - ref[i]));
    printf("%-32s max |C - ref| / max |ref| = %.3e\n", names[mode], worst / den);
  }
  return 0;
}
