// Instruction-cache probe: a loop whose body is KB kilobytes of straight-line 8-byte vector instructions (v_mad_u32_u24, one per 4 cycles
// and wave: the issue density of the strip kernels, ~12,000 instructions per 64,000-cycle block), one 4-wave workgroup per CU on every CU.
// Prints cycles per instruction and wave for each body size: a step above ~64 KB is the cost of a loop body that does not fit the
// instruction cache two CUs share.  Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probes/icache_probe.hip -o /tmp/icache_probe && /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define STR2(x) #x
#define STR(x) STR2(x)

template <int KB>
__global__ __launch_bounds__(256) void body_kernel(unsigned* out, int iters, unsigned b) {
  unsigned a = threadIdx.x;
  for (int i = 0; i < iters; i++) {
    if (KB == 16) asm volatile(".rept 2048\n v_mad_u32_u24 %0, %0, %1, %0\n .endr" : "+v"(a) : "v"(b));
    if (KB == 32) asm volatile(".rept 4096\n v_mad_u32_u24 %0, %0, %1, %0\n .endr" : "+v"(a) : "v"(b));
    if (KB == 48) asm volatile(".rept 6144\n v_mad_u32_u24 %0, %0, %1, %0\n .endr" : "+v"(a) : "v"(b));
    if (KB == 56) asm volatile(".rept 7168\n v_mad_u32_u24 %0, %0, %1, %0\n .endr" : "+v"(a) : "v"(b));
    if (KB == 64) asm volatile(".rept 8192\n v_mad_u32_u24 %0, %0, %1, %0\n .endr" : "+v"(a) : "v"(b));
    if (KB == 80) asm volatile(".rept 10240\n v_mad_u32_u24 %0, %0, %1, %0\n .endr" : "+v"(a) : "v"(b));
    if (KB == 96) asm volatile(".rept 12288\n v_mad_u32_u24 %0, %0, %1, %0\n .endr" : "+v"(a) : "v"(b));
    if (KB == 112) asm volatile(".rept 14336\n v_mad_u32_u24 %0, %0, %1, %0\n .endr" : "+v"(a) : "v"(b));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a;
}

template <int KB>
void run(unsigned* out, int cus, double ghz) {
  const int ninstr = KB * 1024 / 8;
  const int iters = (int)(40000000L / ninstr);       // ~40 M instructions per wave
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(body_kernel<KB>, dim3(cus), dim3(256), 0, 0, out, iters / 8, 3u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(body_kernel<KB>, dim3(cus), dim3(256), 0, 0, out, iters, 3u);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_per = (double)ms * 1e6 / ((double)iters * ninstr);
  printf("body %3d KB (%5d instructions): %.3f ms, %.3f ns per instruction and wave = %.2f cycles at %.2f GHz\n", KB, ninstr, ms, ns_per, ns_per * ghz, ghz);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  unsigned* out;
  hipMalloc(&out, (size_t)cus * 256 * 4);
  printf("%s, %d CUs, nominal %.2f GHz; one 256-thread workgroup per CU\n", p.name, cus, ghz);
  run<16>(out, cus, ghz); run<32>(out, cus, ghz); run<48>(out, cus, ghz); run<56>(out, cus, ghz); run<64>(out, cus, ghz);
  run<80>(out, cus, ghz); run<96>(out, cus, ghz); run<112>(out, cus, ghz);
  hipFree(out);
  return 0;
}
