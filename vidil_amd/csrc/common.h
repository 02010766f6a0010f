// Shared device/host helpers for libvidil_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vidil_hip.h"

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VIDIL_WAVE 64

// error plumbing ------------------------------------------------------------
void vidil_set_error(const char* fmt, ...);

// gemm256.hip: the 256x256 8-wave kernel for large problems (dispatched from vidil_gemm_f16)
bool vidil_gemm256_eligible(const vidil_gemm_args& a, bool any_size = false);
int vidil_gemm256_launch(const vidil_gemm_args& a, hipStream_t s);
const char* vidil_gemm256_variant(const vidil_gemm_args& a);   // "gemm256_kernel" or "gemm4w_kernel": which of the two runs it
bool vidil_gemm4w128_wanted(const vidil_gemm_args& a);          // the 128 x 256-tile form of gemm4w (mid-size grids)
int vidil_gemm4w128_launch(const vidil_gemm_args& a, hipStream_t s);

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

// Fragment-tiled K / V (vidil_gemm_args.kv_tiled, vidil_attention kv_tiled): a (batch, head) owns Tk_cap/32 tiles
// of 32 keys x 64 dims = 2048 halfs, stored in the order the direct attention kernel's MFMA operands want them, so
// that each of its wave-level loads is ONE contiguous KiB (64 lanes x 16 B) instead of 32-byte pieces of 32 rows:
//   K tile: [c/8][key%32][c%8]                      (k-step ks = c/16, half-wave = (c/8)%2, lane = key%32)
//   V tile: [key%32/16][d/32][half-wave][d%32][j]   with the 8 keys of a half-wave in vt_pos order:
//           4-key group g = (key%16)/4 -> half-wave g%2, j = (g/2)*4 + key%4
__host__ __device__ __forceinline__ size_t ktile_off(int t, int c) {
  return (size_t)(t >> 5) * 2048 + (size_t)(((c >> 3) * 32 + (t & 31)) * 8 + (c & 7));
}
__host__ __device__ __forceinline__ size_t vtile_off(int t, int d) {
  const int tt = t & 31, g = (tt & 15) >> 2;
  return (size_t)(t >> 5) * 2048 +
         (size_t)(((((tt >> 4) * 2 + (d >> 5)) * 2 + (g & 1)) * 32 + (d & 31)) * 8 + (g >> 1) * 4 + (tt & 3));
}

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
// f32 -> f16 with the value pinned in a f32 register first: otherwise the compiler may fuse the last
// multiply of an epilogue with the conversion (v_fma_mixlo_f16, ONE rounding) in one kernel and not in
// another (two roundings), and the two disagree on f16 midpoints.
__device__ __forceinline__ f16 to_f16(float v) {
  asm volatile("" : "+v"(v));
  return (f16)v;
}

// The 16-bit operand type of a kernel instantiation (VIDIL_DT_F16 / VIDIL_DT_BF16): vector types, the dense
// v_mfma_f32_32x32x16 of that type, and the pinned f32 -> T rounding (round to nearest even in both cases;
// bf16 through v_cvt_pk_bf16_f32).  Everything else in the kernels — accumulation, softmax, LayerNorm
// statistics, the residual stream — is f32 for both.
template <typename T> struct Elt;
template <> struct Elt<f16> {
  typedef f16x4 x4;
  typedef f16x8 x8;
  static constexpr int kDtype = VIDIL_DT_F16;
  static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f16 from_f32(float v) { return to_f16(v); }
};
template <> struct Elt<bf16> {
  typedef bf16x4 x4;
  typedef bf16x8 x8;
  static constexpr int kDtype = VIDIL_DT_BF16;
  static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ bf16 from_f32(float v) {
    asm volatile("" : "+v"(v));
    return (bf16)v;
  }
};
// fp8 tower mode: OCP e4m3fn bytes (gfx950's native fp8), multiplied 64 k at a time by the block-scaled MFMA with
// the scales fixed at 2^0 (E8M0 code 127): twice the FLOPs of the 16-bit instruction per issue slot.
struct fp8 { uint8_t bits; };
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <> struct Elt<fp8> {
  static constexpr int kDtype = VIDIL_DT_FP8;
};
// four f32 -> four e4m3 bytes (round to nearest even, saturating at +-448 instead of producing NaN)
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f);
  b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f);
  d = fminf(fmaxf(d, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

template <typename T> __device__ __forceinline__ typename Elt<T>::x8 zero8() {
  typename Elt<T>::x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (T)0.f;
  return z;
}
// run `fn(T{})` for the operand type a dtype code names; unknown codes are an argument error
#define VIDIL_DISPATCH_DTYPE(dtype, what, ...)                                         \
  do {                                                                                 \
    if ((dtype) == VIDIL_DT_F16) { using T = f16; __VA_ARGS__; }                       \
    else if ((dtype) == VIDIL_DT_BF16) { using T = bf16; __VA_ARGS__; }                \
    else { vidil_set_error("%s: unknown dtype %d", what, (int)(dtype)); return VIDIL_EINVAL; } \
  } while (0)

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the f16 rounding of the value it
// feeds): 1 v_rcp + 1 v_exp + 5 fma instead of libm's branchy erff.  The GELU epilogue runs on 3072 columns
// of every ViT/MED row and is VALU time the matrix pipe spends idle (one workgroup per CU), so it works on
// PAIRS of values with the packed-f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes'
// worth of IEEE f32 per issue; per component they round exactly like the scalar instructions).
// Every step is an explicit operation (no contraction, no IEEE-division expansion: v_rcp_f32 is 1 ulp and one
// instruction instead of ten) so the instruction sequence, hence every bit of the result, is the same in
// every kernel instantiation: outputs must not depend on which GEMM kernel a batch size selects.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 xs = x * pk_splat(0.70710678118654752440f);
  const f32x2 ax = __builtin_elementwise_abs(xs);
  const f32x2 den = pk_fma(pk_splat(0.3275911f), ax, pk_splat(1.0f));
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 p = pk_fma(pk_splat(1.061405429f), t, pk_splat(-1.453152027f));
  p = pk_fma(p, t, pk_splat(1.421413741f));
  p = pk_fma(p, t, pk_splat(-0.284496736f));
  p = pk_fma(p, t, pk_splat(0.254829592f));
  const f32x2 arg = (ax * ax) * pk_splat(-1.4426950408889634f);   // exp(-x^2) = 2^(-x^2 log2 e)
  const f32x2 e = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  const f32x2 y = pk_fma(-(p * t), e, pk_splat(1.0f));             // erf(|xs|)
  const f32x2 er = {copysignf(y[0], xs[0]), copysignf(y[1], xs[1])};
  return (x * pk_splat(0.5f)) * (er + pk_splat(1.0f));
}
// erf-GELU for 16-BIT (and fp8) OUTPUTS: x * Phi(x) with Phi(x) - 1/2 = x * Q((x / 4.5)^2) on |x| <= 4.5 (Q: degree-8
// weighted least-squares fit on Chebyshev nodes, pinned so that Phi(+-4.5) is exactly 1 / 0; outside, x is clamped, so the
// result is exactly x or 0).  |error| <= 4.8e-5 absolute (rms 2e-5) against erf-GELU over all of f32 — a fifth of an f16
// ulp of the values it rounds to, 1/30 of a bf16 ulp — with NO transcendental: 13 full-rate VALU instructions per value
// instead of ~10 + 2 quarter-rate ones (v_rcp / v_exp): the GELU epilogue of the tower's fc1 GEMM is VALU time the
// matrix pipe idles through.  f32 outputs (the LM-head transform, the parity precision mode) keep
// gelu_erf2.  Explicit operations only, so every kernel instantiation produces the same bits.
__device__ __forceinline__ float gelu_fast1(float x) {
  // plain v_fma_f32 with LITERAL coefficients (v_fmaak_f32): packed-f32 operands must sit in VGPR pairs, and nine
  // splatted constants (18 registers) made the LN-folded GELU instantiation of gemm256 spill 461 registers
  const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
  const float y = xc * (1.0f / 4.5f);
  const float s = y * y;
  float p = __builtin_fmaf(8.050480127e-01f, s, -4.390279192e+00f);
  p = __builtin_fmaf(p, s, 1.052993543e+01f);
  p = __builtin_fmaf(p, s, -1.471975757e+01f);
  p = __builtin_fmaf(p, s, 1.343790172e+01f);
  p = __builtin_fmaf(p, s, -8.530106592e+00f);
  p = __builtin_fmaf(p, s, 3.914417810e+00f);
  p = __builtin_fmaf(p, s, -1.334714149e+00f);
  p = __builtin_fmaf(p, s, 3.986656381e-01f);
  const float phi = __builtin_fmaf(xc, p, 0.5f);
  return x * phi;
}
// (A packed form — v_pk_fma_f32 on pairs, half the VALU cycles: a wave64 plain f32 instruction occupies this SIMD for 4
// cycles, a packed one does two values in the same 4 — needs its nine coefficients in VGPR pairs: the LN-folded GELU
// instantiation of gemm256, already at 247 VGPRs and all 103 SGPRs, then spills 461 registers and runs 2.6x slower
// (measured: 10.2 ms instead of 3.9 ms per launch).  Three more placements of the packed form were compiled — coefficients
// defined opaquely inside the epilogue so that they cannot be hoisted across the main loop, scheduling barriers between
// the pairs, the activation moved to the point where a value is converted for its store instead of updating the
// accumulator tuples in place — all spill 361 registers (fp8 instantiation: 304) in the epilogue region.  The scalar form
// is 52 VALU cycles per value against 62 for the A&S erf with its two quarter-rate transcendentals: +3 % on that kernel
// in situ, 785 -> 811 TFLOP/s.)
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) { return f32x2{gelu_fast1(x[0]), gelu_fast1(x[1])}; }
// The same polynomial on a PAIR with packed-f32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32: two values per 4-cycle issue) —
// the same IEEE operations per element as gelu_fast1, so the results are bit-identical.  For kernels with registers to
// spare for the nine splatted coefficients (gemm4w.hip's epilogue; gemm256's LN-folded GELU instantiation spilled 461
// registers on it).  Only the clamp stays scalar (no packed min / max on gfx950).
// NP pairs are evaluated in LOCKSTEP (every Horner step for all pairs before the next step): a wave alone on its SIMD
// has nobody to fill the dependent-issue gaps of one serial chain (each packed step waits for the previous one), so the
// instruction-level parallelism has to be in the program order.
template <int NP>
__device__ __forceinline__ void gelu_fast2p_n(f32x2 (&x)[NP]) {
  f32x2 xc[NP], s[NP], p[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) xc[i] = f32x2{__builtin_amdgcn_fmed3f(x[i][0], -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x[i][1], -4.5f, 4.5f)};
#pragma unroll
  for (int i = 0; i < NP; ++i) s[i] = xc[i] * pk_splat(1.0f / 4.5f);
#pragma unroll
  for (int i = 0; i < NP; ++i) s[i] = s[i] * s[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) p[i] = pk_fma(pk_splat(8.050480127e-01f), s[i], pk_splat(-4.390279192e+00f));
  constexpr float C[7] = {1.052993543e+01f, -1.471975757e+01f, 1.343790172e+01f, -8.530106592e+00f, 3.914417810e+00f, -1.334714149e+00f, 3.986656381e-01f};
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = pk_fma(p[i], s[i], pk_splat(C[k]));
#pragma unroll
  for (int i = 0; i < NP; ++i) p[i] = pk_fma(xc[i], p[i], pk_splat(0.5f));
#pragma unroll
  for (int i = 0; i < NP; ++i) x[i] = x[i] * p[i];
}
__device__ __forceinline__ f32x2 quick_gelu2(f32x2 x) {
  const f32x2 arg = x * pk_splat(-1.702f * 1.4426950408889634f);   // exp(-1.702 x) as a power of two
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} + pk_splat(1.0f);
  return x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2(f32x2{x, x})[0]; }
__device__ __forceinline__ float quick_gelu(float x) { return quick_gelu2(f32x2{x, x})[0]; }
