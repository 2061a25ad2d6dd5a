// Host-side plumbing shared by the translation units of libtq_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/tq_hip.h"

namespace tq {

int set_error(int code, const char* fmt, ...);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
inline size_t elem_size(int dtype) { return dtype == TQ_F32 ? 4 : 2; }

inline int tuning(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(TQ_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return TQ_OK;
}

int check_quantizer(const tq_quantizer* q, uint64_t n, const char* who);

}  // namespace tq

#define TQ_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return tq::set_error(TQ_EINVAL, __VA_ARGS__);  \
  } while (0)
