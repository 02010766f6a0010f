// Shared device/host helpers for libvidil_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define VIDIL_WAVE 64

// error plumbing ------------------------------------------------------------
void vidil_set_error(const char* fmt, ...);
int vidil_cu_count();                          // core.hip: compute units of the current device (cached; 256 when none answers)
int vidil_lds_opt_in(std::atomic<unsigned long long>& mask, const void* kern, int bytes, const char* who);   // core.hip: per-device dynamic-LDS opt-in, bit set after success
const char* vidil_dev_env(const char* name);   // core.hip: developer overrides, read once per process (live under $VIDIL_DEV_ENV)

// gemm256.hip: the 256x256 8-wave kernel for large problems (dispatched from vidil_gemm_f16)
bool vidil_gemm256_eligible(const vidil_gemm_args& a, bool any_size = false);
int vidil_gemm256_launch(const vidil_gemm_args& a, hipStream_t s);
const char* vidil_gemm256_variant(const vidil_gemm_args& a);   // "gemm256_kernel" or "gemm4w_kernel": which of the two runs it
bool vidil_gemm4w128_wanted(const vidil_gemm_args& a);          // the 128 x 256-tile form of gemm4w (mid-size grids)
int vidil_gemm4w128_launch(const vidil_gemm_args& a, hipStream_t s);
bool vidil_gemm_c3_serves(const vidil_gemm_args& a);            // gemm.hip: a split_k launch that the in-loop compensated kernel takes (at every size)
int vidil_gemm4w_c3_launch(const vidil_gemm_args& a, hipStream_t s);   // gemm4w.hip

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
// erf-GELU for 16-BIT (and fp8) OUTPUTS, transcendental-free: x * Phi(x) with
//     Phi(x) = clamp(1/2 + x * Q(x^2), 0, 1),
// Q a minimax polynomial (absolute error of x * Phi(x), LP fit on [-R, R]; tools/fit_gelu.py prints these coefficients
// and the error tables quoted here).  The clamp is the [0, 1] output modifier of the last fma, so nothing clamps x: past R
// the odd polynomial x * Q(x^2) runs monotonically through +-1/2 (even degree in x^2, positive leading coefficient) and
// the result is exactly x or -0 — checked for every f32 magnitude up to overflow of x^2 (inf * 0 = NaN only for x = -inf).
// Degree by OUTPUT type — the polynomial only has to be invisible under the rounding of the value it feeds:
//   f16  (11 significant bits): degree 8, R = 4.25: |error| <= 4.3e-5 absolute over all of f32 (rms 1.1e-5 on |x| < 4);
//   bf16 ( 8 significant bits): degree 6, R = 4.00: |error| <= 1.9e-4 (rms 1.3e-4): a twentieth of a bf16 ulp at 1.
// These are ABSOLUTE bounds: in the negative tail, where erf-GELU itself is ~1e-4 (x < -3.8), the relative error reaches
// 100 % (the value is then worth 1/10 of an f16 ulp of the fc2 inputs that matter); tests/test_kernels_gpu.py bounds both the
// absolute error and the error in output ulps over every 16-bit input.  Instruction count per PAIR of values (what the
// epilogue of a one-wave-per-SIMD kernel pays for, DESIGN.md §3 "Round 4"): 1 mul + deg fma + 1 fma.clamp + 1 mul = 11 / 9
// packed instructions, against 2 v_med3 + 12 for round 3's clamped degree-8 form.  f32 outputs (the LM-head transform, the
// parity precision mode) keep gelu_erf2.  Explicit IEEE operations only — the scalar and the packed forms below produce
// the same bits in every kernel instantiation.
template <typename TO> struct GeluPoly;
template <> struct GeluPoly<f16> {
  static constexpr int DEG = 8;
  static constexpr float q[9] = {3.988192516e-01f, -6.619034771e-02f, 9.718823738e-03f, -1.079092217e-03f, 8.849158500e-05f,
                                 -5.147754456e-06f, 1.986294828e-07f, -4.515713104e-09f, 4.547582994e-11f};
};
template <> struct GeluPoly<bf16> {
  static constexpr int DEG = 6;
  static constexpr float q[7] = {3.978833846e-01f, -6.457313509e-02f, 8.772395999e-03f, -8.140167550e-04f, 4.795563103e-05f,
                                 -1.598607122e-06f, 2.278152054e-08f};
};
template <typename TO>
__device__ __forceinline__ float gelu_fast1(float x) {
  using P = GeluPoly<TO>;
  const float s = x * x;
  float p = __builtin_fmaf(P::q[P::DEG], s, P::q[P::DEG - 1]);
#pragma unroll
  for (int k = P::DEG - 2; k >= 0; --k) p = __builtin_fmaf(p, s, P::q[k]);
  const float phi = __builtin_amdgcn_fmed3f(__builtin_fmaf(x, p, 0.5f), 0.f, 1.f);   // (-> v_fma_f32 ... clamp)
  return x * phi;
}
template <typename TO>
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) { return f32x2{gelu_fast1<TO>(x[0]), gelu_fast1<TO>(x[1])}; }
// clamp(a * b + 1/2, 0, 1) on a pair: v_pk_fma_f32 with the output modifier (no builtin reaches it for packed f32; the
// scalar form above is folded into `v_fma_f32 ... clamp` by the compiler — the same arithmetic)
__device__ __forceinline__ f32x2 pk_fma_half_clamp(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// The same polynomial on PAIRS with packed-f32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32: two values per issue slot) — the
// same IEEE operations per element as gelu_fast1, so the results are bit-identical.  NP pairs are evaluated in LOCKSTEP
// (every Horner step for all pairs before the next step): a wave alone on its SIMD has nobody to fill the dependent-issue
// gaps of one serial chain, so the instruction-level parallelism has to be in the program order.
template <typename TO, int NP>
__device__ __forceinline__ void gelu_fast2p_n(f32x2 (&x)[NP]) {
  using P = GeluPoly<TO>;
  f32x2 s[NP], p[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) s[i] = x[i] * x[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) p[i] = pk_fma(pk_splat(P::q[P::DEG]), s[i], pk_splat(P::q[P::DEG - 1]));
#pragma unroll
  for (int k = P::DEG - 2; k >= 0; --k)
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = pk_fma(p[i], s[i], pk_splat(P::q[k]));
#pragma unroll
  for (int i = 0; i < NP; ++i) p[i] = pk_fma_half_clamp(x[i], p[i]);
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
