// Host-side helpers for the C ABI (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

void hftt_set_error(const char* fmt, ...);

#define HFTT_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      hftt_set_error(__VA_ARGS__);         \
      return 1;                            \
    }                                      \
  } while (0)

// One process drives ONE device (one process per GPU: DESIGN.md section 7).  The launchers cache per-kernel state in process-wide statics --
// the dynamic-LDS attribute already set, the CU count, the occupancy-derived grid of the persistent kernels -- which belong to the device of the
// first launch.  A launch on another device fails loudly here instead of skipping the attribute call and reusing the first device's grid
// (ADVICE r05).  0 = ok.
int hftt_device_guard(const char* what);

#define HFTT_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      hftt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return 2;                                                              \
    }                                                                        \
    if (hftt_device_guard(name) != 0) return 3;                              \
  } while (0)

static inline int hftt_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
