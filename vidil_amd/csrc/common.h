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

// gemm256.hip: the 256x256 8-wave kernel for large problems (dispatched from vidil_gemm_f16)
bool vidil_gemm256_eligible(const vidil_gemm_args& a);
int vidil_gemm256_launch(const vidil_gemm_args& a, hipStream_t s);
// gemm256w4.hip: same tile on 4 waves (one per SIMD, 512 registers each)
int vidil_gemm256w4_launch(const vidil_gemm_args& a, hipStream_t s);

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

// V^T key order.  The transposed-score MFMA layout leaves a half-wave holding keys {0-3, 8-11} (lanes
// 0-31) or {4-7, 12-15} (lanes 32-63) of every 16-key block, so V^T rows store the keys of a block in the
// order 0-3, 8-11, 4-7, 12-15: each half-wave's 8 keys are then 16 contiguous bytes (one load instead of
// two).  vt_pos maps key -> storage column (and back: it is an involution); row strides are multiples of 16.
__host__ __device__ __forceinline__ int vt_pos(int t) { return t ^ ((((t >> 2) ^ (t >> 3)) & 1) * 12); }

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
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the f16 rounding of the value it
// feeds): 1 rcp + 1 exp + 5 fma instead of libm's branchy erff — the GELU epilogue runs on 3072 columns
// of every ViT/MED row and was VALU-bound with erff.
// f32 -> f16 with the value pinned in a f32 register first: otherwise the compiler may fuse the last
// multiply of an epilogue with the conversion (v_fma_mixlo_f16, ONE rounding) in one kernel and not in
// another (two roundings), and the two disagree on f16 midpoints.
__device__ __forceinline__ f16 to_f16(float v) {
  asm volatile("" : "+v"(v));
  return (f16)v;
}

// Every step is an explicit correctly-rounded intrinsic so the instruction sequence (hence every bit of the
// result) is the same in every kernel instantiation: outputs must not depend on which GEMM kernel a batch
// size selects.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(__fmaf_rn(0.3275911f, ax, 1.0f));
  float p = __fmaf_rn(1.061405429f, t, -1.453152027f);
  p = __fmaf_rn(p, t, 1.421413741f);
  p = __fmaf_rn(p, t, -0.284496736f);
  p = __fmaf_rn(p, t, 0.254829592f);
  const float e = __expf(-__fmul_rn(ax, ax));
  const float y = __fsub_rn(1.0f, __fmul_rn(__fmul_rn(p, t), e));
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float e = erf_as(__fmul_rn(x, 0.70710678118654752440f));
  return __fmul_rn(__fmul_rn(0.5f, x), __fadd_rn(1.0f, e));
}
__device__ __forceinline__ float quick_gelu(float x) {
  const float d = __fadd_rn(1.0f, __expf(__fmul_rn(-1.702f, x)));
  return __fdiv_rn(x, d);
}
