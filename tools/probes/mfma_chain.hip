// GPU box probe: does the ORDER of independent accumulations matter?  Eight accumulators, 128 v_mfma_f32_32x32x16_bf16 per iteration, issued as
// runs of CH consecutive instructions into the same accumulator (CH = 1: round-robin; CH = 16: one accumulator's whole k range, then the next).
// Cycles per instruction, the shader clock the loop is granted (s_memtime / s_memrealtime) and the resulting nanoseconds per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, int iters, int seed) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
  for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  bf16x8 ab[4], bb[4];
  for (int q = 0; q < 4; q++) for (int e = 0; e < 8; e++) { ab[q][e] = (__bf16)(0.37f * ((seed * 7 + e * 13 + q * 5 + lane * 3) % 17 - 8)); bb[q][e] = (__bf16)(0.21f * ((seed + e * 5 + q * 11 + lane * 7) % 19 - 9)); }
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 128; i++) {
      const int a = (i / CH) % 8;
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[i & 3], bb[(i >> 2) & 3], acc[a], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) { out[1 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[2 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = r1 - r0; }
}
template <int CH>
void run() {
  unsigned long long* d; (void)hipMalloc(&d, 8 * (1 + 256 * 4 * 2));
  const int iters = 3000;
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((probe<CH>), dim3(256), dim3(256), 0, 0, d, iters, 3);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(1 + 256 * 4 * 2);
  (void)hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  double ticks = 0, rt = 0;
  for (int i = 0; i < 1024; i++) { ticks += (double)h[1 + 2 * i]; rt += (double)h[2 + 2 * i]; }
  const double per = ticks / 1024 / (iters * 128.0), ghz = ticks / rt * 0.1;
  printf("runs of %2d into one accumulator: %5.1f cycles / instruction at %.2f GHz = %5.2f ns\n", CH, per, ghz, per / ghz);
  (void)hipFree(d);
}
int main() { for (int round = 0; round < 3; round++) { printf("-- round %d\n", round); run<1>(); run<2>(); run<4>(); run<16>(); } return 0; }
