// GPU box probe: cycles per v_mfma_f32_32x32x16_bf16 as issued by ONE wave per SIMD (256-thread workgroups, one per CU), for the shapes the
// strip kernels use: a single accumulation chain (the first GEMM of the fused FFN), eight accumulators round-robin (the second GEMM), each with
// register operands and with the A fragment re-read from LDS every MFMA (one ds_read_b128 per MFMA, six reads in flight).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int AHEAD>
__global__ __launch_bounds__(256, 1) void probe_pipe(unsigned long long* out, int iters, float seed) {
  // the slot loop of the strip kernels: sixteen fragments of a 16 KB slot, AHEAD reads in flight (sched_group_barrier pins the shape)
  __shared__ __attribute__((aligned(16))) unsigned char sm[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(sm)[i] = seed * (i & 255);
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  bf16x8 b; for (int e = 0; e < 8; e++) b[e] = (__bf16)(seed + e + lane);
  const unsigned char* base = sm + lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    const unsigned char* slot = base + (it & 3) * 16384;
    bf16x8 fr[16];
#pragma unroll
    for (int i = 0; i < 16; i++) fr[i] = *reinterpret_cast<const bf16x8*>(slot + i * 1024);
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], b, acc[i % NACC], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
    for (int k = 0; k < 16 - AHEAD; k++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, bool LDS>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(sm)[i] = seed * (i & 255);
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  bf16x8 b; for (int e = 0; e < 8; e++) b[e] = (__bf16)(seed + e + lane);
  bf16x8 a0; for (int e = 0; e < 8; e++) a0[e] = (__bf16)(seed * 2 + e);
  const unsigned char* base = sm + lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      bf16x8 a = a0;
      if (LDS) a = *reinterpret_cast<const bf16x8*>(base + ((it & 3) * 16 + i) * 1024);
      acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NACC], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, bool LDS>
void run(const char* name) {
  unsigned long long* d; hipMalloc(&d, 8 * (1 + 256 * 4));
  const int iters = 2000;
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((probe<NACC, LDS>), dim3(256), dim3(256), 0, 0, d, iters, 0.001f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1 + 256 * 4);
  hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 1; i < (int)h.size(); i++) sum += (double)h[i];
  printf("%-52s %6.1f cycles per MFMA\n", name, sum / (256.0 * 4) / (iters * 16.0));
  hipFree(d);
}
template <int NACC, int AHEAD>
void run_pipe(const char* name) {
  unsigned long long* d; hipMalloc(&d, 8 * (1 + 256 * 4));
  const int iters = 2000;
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((probe_pipe<NACC, AHEAD>), dim3(256), dim3(256), 0, 0, d, iters, 0.001f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1 + 256 * 4);
  hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 1; i < (int)h.size(); i++) sum += (double)h[i];
  printf("%-52s %6.1f cycles per MFMA\n", name, sum / (256.0 * 4) / (iters * 16.0));
  hipFree(d);
}
int main() {
  run_pipe<1, 2>("slot loop, one chain, 2 reads in flight");
  run_pipe<1, 6>("slot loop, one chain, 6 reads in flight");
  run_pipe<8, 6>("slot loop, eight accumulators, 6 reads in flight");
  run_pipe<8, 12>("slot loop, eight accumulators, 12 reads in flight");
  run<1, false>("one accumulation chain, register operands");
  run<8, false>("eight accumulators, register operands");
  run<1, true>("one accumulation chain, A fragment from LDS");
  run<8, true>("eight accumulators, A fragment from LDS");
  return 0;
}
