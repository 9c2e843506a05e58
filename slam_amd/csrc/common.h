// Shared helpers for the libslamhip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/slamhip.h"

// hipGetLastError() is sticky per thread: an error left behind by unrelated HIP
// calls of the host application would otherwise be reported as ours.
#define SLAMHIP_BEGIN() ((void)hipGetLastError())

#define SLAMHIP_CHECK_LAUNCH()                         \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) return SLAMHIP_ELAUNCH;     \
  } while (0)

static inline int slamhip_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t slamhip_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Number of 32-bit words a packed descriptor row occupies: 8, 16 or 32
// (32-, 64- or 128-byte rows; shorter descriptors are zero padded, which is
// neutral under XOR + popcount).
static inline int slamhip_packed_words(int desc_bytes) {
  if (desc_bytes <= 32) return 8;
  if (desc_bytes <= 64) return 16;
  return 32;
}
