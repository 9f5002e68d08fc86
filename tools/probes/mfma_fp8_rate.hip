// GPU box probe: issue rate of the block-scaled fp8 MFMA of gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4, both operands e4m3) against the bf16 MFMA the
// x3 kernels use (v_mfma_f32_32x32x16_bf16), from registers, ONE wave per SIMD, eight accumulators round-robin; also the realtime counter so
// the shader clock under each loop is known.  65,536 multiply-adds per fp8 instruction against 16,384 per bf16 instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_fp8_rate.hip -o /tmp/mfma_fp8_rate && /tmp/mfma_fp8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// one x3 product over 64 k-elements: THREE = twelve bf16 instructions (hi.hi, hi.lo, lo.hi), MIX = four bf16 (hi.hi) + two fp8 (the cross terms)
template <bool MIX>
__global__ __launch_bounds__(256, 1) void probe_x3(unsigned long long* out, int iters, int seed) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
  for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  bf16x8 ab, bb; for (int e = 0; e < 8; e++) { ab[e] = (__bf16)(0.001f * (seed + e)); bb[e] = (__bf16)(0.002f * (seed + e + lane)); }
  i32x8 a8, b8; for (int e = 0; e < 8; e++) { a8[e] = 0x38383838 + seed * (e + 1); b8[e] = 0x30303030 + (seed + lane) * (e + 3); }
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 8; t++) {                      // eight output tiles, each its 64 k-elements
      if (MIX) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 127, 0, 127);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc[t], 0, 0, 0, 127, 0, 127);
      } else {
#pragma unroll
        for (int i = 0; i < 12; i++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) { out[1 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[2 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = r1 - r0; }
}
template <bool MIX>
void run_x3(const char* name) {
  unsigned long long* d; hipMalloc(&d, 8 * (1 + 256 * 4 * 2));
  const int iters = 5000;
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((probe_x3<MIX>), dim3(256), dim3(256), 0, 0, d, iters, 3);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1 + 256 * 4 * 2);
  hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  double ticks = 0, rt = 0;
  for (int i = 0; i < 1024; i++) { ticks += (double)h[1 + 2 * i]; rt += (double)h[2 + 2 * i]; }
  const double per = ticks / 1024 / (iters * 8.0), ghz = ticks / rt * 0.1;
  printf("%-46s %6.1f cycles per (tile, 64 k) at %.2f GHz = %.1f ns\n", name, per, ghz, per / ghz);
  hipFree(d);
}

template <bool FP8>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, int iters, int seed) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
  for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  bf16x8 ab, bb; for (int e = 0; e < 8; e++) { ab[e] = (__bf16)(0.001f * (seed + e)); bb[e] = (__bf16)(0.002f * (seed + e + lane)); }
  i32x8 a8, b8; for (int e = 0; e < 8; e++) { a8[e] = 0x38383838 + seed * (e + 1); b8[e] = 0x30303030 + (seed + lane) * (e + 3); }
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (FP8) acc[i % 8] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i % 8], 0, 0, 0, 127, 0, 127);   // cbsz 0 / blgp 0: e4m3 x e4m3; scales 2^0
      else acc[i % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i % 8], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) { out[1 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[2 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = r1 - r0; }
}

template <bool FP8>
void run(const char* name, double macs) {
  unsigned long long* d; hipMalloc(&d, 8 * (1 + 256 * 4 * 2));
  const int iters = 20000;
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((probe<FP8>), dim3(256), dim3(256), 0, 0, d, iters, 3);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1 + 256 * 4 * 2);
  hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  double ticks = 0, rt = 0;
  for (int i = 0; i < 1024; i++) { ticks += (double)h[1 + 2 * i]; rt += (double)h[2 + 2 * i]; }
  ticks /= 1024; rt /= 1024;
  const double per = ticks / (iters * 16.0), ghz = ticks / rt * 0.1;
  // 4 SIMDs x 256 CUs issue one instruction per `per` cycles each
  printf("%-34s %6.1f cycles / instruction at %.2f GHz  ->  %.0f TFLOP/s dense (2 x MACs) over 1024 SIMDs\n", name, per, ghz, 2.0 * macs / per * ghz * 1e9 * 1024 / 1e12);
  hipFree(d);
}
int main() {
  // several rounds: the clock a loop is granted also depends on what ran in the seconds before it (the first kernels of a process run slower)
  for (int round = 0; round < 4; round++) {
    printf("-- round %d\n", round);
    run<false>("v_mfma_f32_32x32x16_bf16", 32.0 * 32 * 16);
    run<true>("v_mfma_scale_f32_32x32x64_f8f6f4", 32.0 * 32 * 64);
    run_x3<false>("x3 product, three bf16 passes (12 instr.)");
    run_x3<true>("x3 product, bf16 hi.hi + fp8 cross (4 + 2 instr.)");
  }
  return 0;
}
