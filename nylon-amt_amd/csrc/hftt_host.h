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

#define HFTT_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      hftt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return 2;                                                              \
    }                                                                        \
  } while (0)

static inline int hftt_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
