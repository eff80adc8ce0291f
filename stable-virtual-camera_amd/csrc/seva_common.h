// Shared device/host helpers for libseva_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/seva_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SEVA_WAVE 64

// host side ----------------------------------------------------------------------------------
void seva_set_error(const char* fmt, ...);
int seva_check_launch(const char* what);

#define SEVA_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      seva_set_error(__VA_ARGS__);       \
      return SEVA_ERR_ARG;               \
    }                                    \
  } while (0)

// profiling (capi.hip) -- brackets a launch with events when enabled
struct SevaProfScope {
  int cls;
  double work;
  hipStream_t stream;
  hipEvent_t e0, e1;
  bool on;
  SevaProfScope(int cls, double work, hipStream_t stream);
  ~SevaProfScope();
};

// device side --------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
