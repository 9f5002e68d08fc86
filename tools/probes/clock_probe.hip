// dev probe (built on the GPU box by tools/clock_trace.sh, not part of libhftt_hip.so).  clock_sample: ONE wave reads the constant 100 MHz
// counter (s_memrealtime) and the shader-clock counter (s_memtime) twice, `window` realtime ticks apart, and writes (start, d realtime,
// d shader) -- the shader clock at that moment.  (The shader counter is only comparable within one wave: samples of different CUs differ by
// arbitrary offsets.)  Enqueued on the step's stream behind every plan entry (tools/clock_trace.py) it records the clock the power management
// is granting right behind each launch; the realtime stamps of consecutive samples give the launch durations.
#include <hip/hip_runtime.h>
extern "C" __global__ void clock_sample_kernel(long long* out, int idx, int window) {
  if (threadIdx.x != 0) return;
  const long long rt0 = (long long)__builtin_amdgcn_s_memrealtime();
  const long long sc0 = (long long)__builtin_amdgcn_s_memtime();
  long long rt1;
  int guard = 0;
  do { __builtin_amdgcn_s_sleep(2); rt1 = (long long)__builtin_amdgcn_s_memrealtime(); } while (rt1 < rt0 + window && ++guard < 100000);
  const long long sc1 = (long long)__builtin_amdgcn_s_memtime();
  out[3 * idx] = rt0; out[3 * idx + 1] = rt1 - rt0; out[3 * idx + 2] = sc1 - sc0;
}
// clock_trace: ONE wave on a side stream, resident for the whole measurement: every `period` realtime ticks it appends (realtime, shader
// counter); it leaves after n samples (bounded).  The shader clock between two samples is the clock of the XCD the wave sits on while the
// step's kernels run beside it.
extern "C" __global__ void clock_trace_kernel(long long* out, int n, int period) {
  if (threadIdx.x != 0) return;
  long long next = (long long)__builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; i++) {
    long long rt;
    int guard = 0;
    do { __builtin_amdgcn_s_sleep(16); rt = (long long)__builtin_amdgcn_s_memrealtime(); } while (rt < next && ++guard < 1000000);
    out[2 * i] = rt;
    out[2 * i + 1] = (long long)__builtin_amdgcn_s_memtime();
    next = rt + period;
  }
}
extern "C" int clock_trace(long long* out, int n, int period, void* stream) {
  hipLaunchKernelGGL(clock_trace_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out, n, period);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" int clock_sample(long long* out, int idx, int window, void* stream) {
  hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out, idx, window);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
