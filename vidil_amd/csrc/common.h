// Shared device/host helpers for libvidil_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vidil_hip.h"

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VIDIL_WAVE 64

// error plumbing ------------------------------------------------------------
void vidil_set_error(const char* fmt, ...);

#define VIDIL_REQUIRE(cond, ...)                \
  do {                                          \
    if (!(cond)) {                              \
      vidil_set_error(__VA_ARGS__);             \
      return VIDIL_EINVAL;                      \
    }                                           \
  } while (0)

#define VIDIL_CHECK_LAUNCH(what)                                            \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      vidil_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
      return VIDIL_ELAUNCH;                                                 \
    }                                                                       \
  } while (0)

// device helpers --------------------------------------------------------------
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
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float quick_gelu(float x) {
  return x / (1.0f + __expf(-1.702f * x));
}
