// Shared host/device helpers of libpaged_attn_mi355x (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "paged_attn_mi355x.h"

namespace pa {

// ---- error reporting ------------------------------------------------------

// printf-style; stores the message for pa_last_error() and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

#define PA_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return ::pa::fail(PA_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
  } while (0)

// Checks the launch itself (invalid configuration, missing code object);
// execution errors surface asynchronously, as with any HIP launch.
#define PA_CHECK_LAUNCH(what)                                                      \
  do {                                                                             \
    hipError_t pa_e_ = hipGetLastError();                                          \
    if (pa_e_ != hipSuccess)                                                       \
      return ::pa::fail(PA_ERR_HIP, "%s: %s", what, hipGetErrorString(pa_e_));     \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int elem_size(int dtype) { return dtype == PA_DTYPE_F32 ? 4 : 2; }

// A launch-form constant that a tuning build (-DPA_TUNING, the Makefile's
// `variant` target) lets an environment variable override for A/B runs.  The
// product build never reads the environment: its launch forms are fixed.
inline int tuning_int(const char* env_name, int value) {
#ifdef PA_TUNING
  const char* e = getenv(env_name);
  if (e != nullptr) return atoi(e);
#else
  (void)env_name;
#endif
  return value;
}

// ---- device-side vector types ---------------------------------------------

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// A workgroup index permutation that keeps a contiguous range of work items
// on the workgroups that share an XCD.  Workgroups are dealt round-robin over
// the 8 XCDs, so ids b and b + 8 share one L2; work item w = f(b) is chosen
// such that the ids of one XCD walk a contiguous slice of [0, n).  Bijective
// for every n.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned n) {
  const unsigned xcd = b & 7u, k = b >> 3;
  const unsigned q = n >> 3, r = n & 7u;
  const unsigned start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return start + k;
}

}  // namespace pa
