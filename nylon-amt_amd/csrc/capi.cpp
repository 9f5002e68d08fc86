// C ABI plumbing: version, thread-local error string, device query.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <atomic>
#include <stdio.h>
#include "../../include/hftt_hip.h"
#include "hftt_host.h"

static thread_local char g_err[512] = "";

void hftt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hftt_device_guard(const char* what) {
  static std::atomic<int> first{-1};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;      // (the launch check in front of this has already reported a broken runtime)
  int seen = -1;
  if (first.compare_exchange_strong(seen, dev) || seen == dev) return 0;
  hftt_set_error("%s: this process launched on device %d before and is now on device %d -- one process drives one device (the launch "
                 "caches of libhftt_hip.so are per process: LDS attributes, CU count, resident-workgroup grids); start one process per GPU", what, seen, dev);
  return 3;
}

extern "C" int hftt_abi_version(void) { return HFTT_ABI_VERSION; }
extern "C" const char* hftt_last_error(void) { return g_err; }
// bit 0: built with the opt-in gradient-rounding forms (HFTT_BUILD_GRAD_HI=1: HFTT_SL_X3_GRAD_HI / HFTT_TN_DY_HI / HFTT_NT_A_HI)
extern "C" int hftt_build_options(void) {
#ifdef HFTT_GRAD_HI_BUILD
  return 1;
#else
  return 0;
#endif
}
extern "C" int hftt_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
  return p.multiProcessorCount;
}
