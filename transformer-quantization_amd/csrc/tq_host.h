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

// Row-parameter layouts ([outer, n_params, inner]: per-token ranges, per-channel weights) with short rows: one wave per
// (parameter, slice of the outer index) -- mm_rows_wave (tq_stats.hip), fq_rows_wave (tq_fake_quant.hip)
constexpr uint64_t kWaveRowMaxVec = 512;    // rows of up to 512 16-byte vectors (8 KB)
constexpr uint64_t kWaveRowTarget = 8192;   // waves wanted in flight: 256 CUs x 4 SIMDs x 8

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

// Sharded per-tensor calibration, second half as ONE launch (tq_fake_quant.hip): every block derives the estimator
// update and the quantizer parameters from the all-reduced statistics stats = [-min, max] in registers and quantizes its
// tile with them; block 0 also writes the new state.  Requires 16-byte aligned x / y, and that no block reads memory
// block 0 writes: cur_* / delta / zero_float / signed_flag must not alias prev_* (the single-GPU step hands over a COPY
// of the previous state that its statistics launch made, so in-place state is fine there).
struct CalibApplyArgs {
  const float* stats;             // [-min, max] (sharded step: all-reduced), or NULL with `partials`
  const float* partials;          // single-GPU step: per-block (min, max) pairs of the statistics launch, [n_partials][2]
  uint32_t n_partials;
  const float *prev_min, *prev_max;
  float *cur_min, *cur_max, *delta, *zero_float;
  uint8_t* signed_flag;
  int mode, n_bits, symmetric, log_domain;
  float eps, om, mom;
};
int launch_fq_from_stats(const void* x, void* y, uint64_t n, int dtype, const CalibApplyArgs& c, hipStream_t st);

// Dynamic / calibrating step for row-parameter layouts whose per-parameter data fits in one block's registers (per-token
// ranges at inference batch sizes: [8, 128, 768] -> 8 rows x 768 per token position): statistics, estimator rule, range ->
// parameters and the fake-quant itself in ONE launch with ONE read of x (tq_fake_quant.hip; arithmetic of calib_update_k +
// make_qp, so the result is bit-identical to the statistics / update / quantize launches).  -> -1 when the shape does not
// qualify (the caller runs the separate launches), else a TQ_* code.
struct RowsOnePassArgs {
  uint64_t outer, n_params, inner;
  int mode, n_bits, log_domain;
  const float *prev_min, *prev_max;
  float *cur_min, *cur_max, *delta, *zero_float;
  float eps, om, mom;
};
int launch_calib_rows_onepass(const void* x, void* y, int dtype, const RowsOnePassArgs& a, hipStream_t st);

// halves of the one-call sharded steps (tq_stats.hip); `stats` scratch of >= 4 floats
int calibrate_stats_for_exchange(const void* x, uint64_t n, int dtype, uint64_t n_params, uint64_t inner, float* stats,
                                 void* workspace, size_t workspace_bytes, uint32_t* counter, const float* prev_min,
                                 const float* prev_max, int* prev_in_stats, tq_stream_t stream);
int calibrate_apply_after_exchange(const float* stats, const void* x, uint64_t n, int dtype, uint64_t n_params,
                                   uint64_t inner, int mode, const float* prev_min, const float* prev_max, float* cur_min,
                                   float* cur_max, double momentum, uint64_t n_groups, const int64_t* order, int n_bits,
                                   int symmetric, float eps, int log_domain, float* delta, float* zero_float,
                                   uint8_t* signed_flag, void* y, int prev_in_stats, tq_stream_t stream);

}  // namespace tq

#define TQ_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return tq::set_error(TQ_EINVAL, __VA_ARGS__);  \
  } while (0)
