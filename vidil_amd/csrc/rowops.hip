// rowops.hip — HBM-bound row kernels: LayerNorm, patch extraction (im2col for
// stride==kernel conv, fused with uint8 -> normalised f16), CLS row, token
// embedding, row gather, L2 normalisation.  All are one wave per row (or per
// patch), 16-B vector accesses, f32 statistics.
#include <type_traits>

#include "common.h"

namespace {

// One wave per row; D/64 elements per lane, processed as float4 where D%256==0
// is not required: lane handles elements lane*4 + 256*i (+0..3).
template <typename T, int VPL>  // float4 vectors per lane: D = 256*VPL
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t x_stride,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int M,
                                                        T* __restrict__ out16, float* __restrict__ out32, int split3) {
  using x4 = typename std::conditional<sizeof(T) == 1, uint32_t, typename Elt<typename std::conditional<sizeof(T) == 1, f16, T>::type>::x4>::type;
  constexpr int D = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * x_stride;
  f32x4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = __builtin_nontemporal_load((const f32x4*)(xr + i * 256 + lane * 4));   // streamed once: do not keep in L2
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      ss += d * d;
    }
  }
  const float var = wave_sum(ss) * (1.0f / D);
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = i * 256 + lane * 4;
    const f32x4 g = *(const f32x4*)(gamma + c);
    const f32x4 b = *(const f32x4*)(beta + c);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
    if (out32 != nullptr) *(f32x4*)(out32 + (size_t)row * D + c) = y;
    if (out16 != nullptr) {
      if constexpr (sizeof(T) == 1) {
        __builtin_nontemporal_store(pack4_fp8(y[0], y[1], y[2], y[3]), (uint32_t*)((uint8_t*)out16 + (size_t)row * D + c));
      } else if (split3) {      // error-compensated operand rows [hi | lo | hi], row stride 3D (VIDIL_DT_SPLIT3)
        x4 h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h4[e] = Elt<T>::from_f32(y[e]);
          l4[e] = Elt<T>::from_f32(y[e] - (float)h4[e]);
        }
        T* o = out16 + (size_t)row * 3 * D + c;
        *(x4*)o = h4;
        *(x4*)(o + D) = l4;
        if (split3 != 2) *(x4*)(o + 2 * D) = h4;        // (2 = VIDIL_DT_SPLIT2: planes hi | lo only)
      } else {
        __builtin_nontemporal_store(x4{(T)y[0], (T)y[1], (T)y[2], (T)y[3]}, (x4*)(out16 + (size_t)row * D + c));
      }
    }
  }
}

// out[m][0:D] = hi, out[m][D:2D] = lo, out[m][2D:3D] = hi   with hi = T(x), lo = T(x - hi)
template <typename T>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, T* __restrict__ out, int M, int D) {
  using x4 = typename Elt<T>::x4;
  const size_t total = (size_t)M * (D / 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / (D / 4);
    const int c = (int)(i - m * (D / 4)) * 4;
    const f32x4 v = *(const f32x4*)(x + m * D + c);
    x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = Elt<T>::from_f32(v[e]);
      lo[e] = Elt<T>::from_f32(v[e] - (float)hi[e]);
    }
    T* o = out + m * 3 * D + c;
    *(x4*)o = hi;
    *(x4*)(o + D) = lo;
    *(x4*)(o + 2 * D) = hi;
  }
}

// out[(b*P + py*G + px), c*ps*ps + y*ps + x] = img[b, c, py*ps + y, px*ps + x]
// one thread = 8 consecutive x of one (patch, c, y) line.
template <typename T>
__global__ __launch_bounds__(256) void patchify_f32_kernel(const float* __restrict__ img, T* __restrict__ out,
                                                           int B, int S, int ps) {
  using x8 = typename Elt<T>::x8;
  const int G = S / ps;
  const int xch = ps / 8;                      // 8-pixel chunks per patch line
  const size_t total = (size_t)B * G * G * 3 * ps * xch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int xc = r % xch; r /= xch;
    const int y = r % ps; r /= ps;
    const int c = r % 3; r /= 3;
    const int px = r % G; r /= G;
    const int py = r % G; r /= G;
    const int b = (int)r;
    const float* src = img + (((size_t)b * 3 + c) * S + (py * ps + y)) * S + px * ps + xc * 8;
    const f32x4 a0 = *(const f32x4*)(src), a1 = *(const f32x4*)(src + 4);
    x8 o = {(T)a0[0], (T)a0[1], (T)a0[2], (T)a0[3], (T)a1[0], (T)a1[1], (T)a1[2], (T)a1[3]};
    T* dst = out + ((size_t)(b * G + py) * G + px) * (3 * ps * ps) + (c * ps + y) * ps + xc * 8;
    *(x8*)dst = o;
  }
}

struct Norm3 { float scale[3]; float shift[3]; };  // y = x*scale + shift

// uint8 HWC -> normalised f16 patch rows.  One thread = 8 consecutive pixels
// (24 B read, three 16-B stores to the three channel planes of the patch row).
template <typename T>
__global__ __launch_bounds__(256) void patchify_u8_kernel(const uint8_t* __restrict__ img, T* __restrict__ out,
                                                          int B, int S, int ps, Norm3 nm) {
  using x8 = typename Elt<T>::x8;
  const int G = S / ps;
  const int xch = ps / 8;
  const size_t total = (size_t)B * G * G * ps * xch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int xc = r % xch; r /= xch;
    const int px = r % G; r /= G;       // px before y: consecutive threads walk a frame row
    const int y = r % ps; r /= ps;
    const int py = r % G; r /= G;
    const int b = (int)r;
    const uint8_t* src = img + (((size_t)b * S + (py * ps + y)) * S + px * ps + xc * 8) * 3;
    // 24 bytes, 8-B aligned (pixel index multiple of 8)
    const uint2 w0 = *(const uint2*)(src), w1 = *(const uint2*)(src + 8), w2 = *(const uint2*)(src + 16);
    uint8_t bytes[24];
    *(uint2*)(bytes) = w0; *(uint2*)(bytes + 8) = w1; *(uint2*)(bytes + 16) = w2;
    T* dst = out + ((size_t)(b * G + py) * G + px) * (3 * ps * ps) + y * ps + xc * 8;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (T)((float)bytes[e * 3 + c] * nm.scale[c] + nm.shift[c]);
      *(x8*)(dst + c * ps * ps) = o;
    }
  }
}

// Any patch size (CLIP ViT-L/14: ps = 14, 588 columns): one thread per output element; rows are ldk =
// round_up(3*ps*ps, 64) halfs long and zero padded so the patch-embedding GEMM keeps its K % 64 == 0 contract
// (the weight is zero padded the same way by the host packing: exact).
template <typename T, bool U8>
__global__ __launch_bounds__(256) void patchify_any_kernel(const void* __restrict__ img, T* __restrict__ out, int B, int S,
                                                           int ps, int ldk, Norm3 nm, int split3 = 0) {
  const int G = S / ps;
  const int pp = ps * ps;
  const size_t total = (size_t)B * G * G * ldk;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % ldk);
    size_t r = i / ldk;
    const int px = r % G; r /= G;
    const int py = r % G; r /= G;
    const int b = (int)r;
    float v = 0.f;
    if (col < 3 * pp) {
      const int c = col / pp, rem = col - c * pp;
      const int y = rem / ps, x = rem - y * ps;
      const size_t row = py * ps + y, cx = px * ps + x;
      if constexpr (U8) {
        v = (float)((const uint8_t*)img)[(((size_t)b * S + row) * S + cx) * 3 + c] * nm.scale[c] + nm.shift[c];
      } else {
        v = ((const float*)img)[(((size_t)b * 3 + c) * S + row) * S + cx];
      }
    }
    if (split3) {       // error-compensated operand rows [hi | lo | hi], 3*ldk wide (VIDIL_DT_SPLIT3)
      const T h = Elt<T>::from_f32(v);
      T* o = out + (i / ldk) * (size_t)(3 * ldk) + col;
      o[0] = h;
      o[ldk] = Elt<T>::from_f32(v - (float)h);
      o[2 * ldk] = h;
    } else {
      out[i] = (T)v;
    }
  }
}

__global__ void set_cls_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos0,
                               int B, int T, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  x[(size_t)b * T * D + d] = cls[d] + pos0[d];
}

__global__ __launch_bounds__(256) void embed_tokens_kernel(const int32_t* __restrict__ ids, const float* __restrict__ word,
                                                          const float* __restrict__ pos, float* __restrict__ out,
                                                          int M, int T, int pos_off, int D, int vocab) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* w = word + (size_t)id * D;
  const float* pe = pos + (size_t)(pos_off + row % T) * D;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *(const f32x4*)(w + c), b = *(const f32x4*)(pe + c);
    *(f32x4*)(out + (size_t)row * D + c) = a + b;
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int n, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* src = x + (size_t)idx[row] * D;
  for (int c = lane * 4; c < D; c += 256) *(f32x4*)(out + (size_t)row * D + c) = *(const f32x4*)(src + c);
}

__global__ __launch_bounds__(256) void l2norm_kernel(float* __restrict__ x, int n, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  float* xr = x + (size_t)row * D;
  float ss = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *(const f32x4*)(xr + c);
    ss += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  }
  const float nrm = sqrtf(wave_sum(ss));
  for (int c = lane * 4; c < D; c += 256) {
    f32x4 a = *(const f32x4*)(xr + c);
    a[0] /= nrm; a[1] /= nrm; a[2] /= nrm; a[3] /= nrm;
    *(f32x4*)(xr + c) = a;
  }
}

inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  return (int)(g > 256 * 16 ? 256 * 16 : (g == 0 ? 1 : g));
}

}  // namespace

template <typename T>
static int layernorm_launch(const float* x, int64_t x_stride, const float* gamma, const float* beta, float eps, int M, int D,
                            T* out16, float* out32, hipStream_t s, int split3 = 0) {
  dim3 grid((M + 3) / 4), block(256);
  switch (D) {
    case 256: hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, block, 0, s, x, x_stride, gamma, beta, eps, M, out16, out32, split3); break;
    case 512: hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, s, x, x_stride, gamma, beta, eps, M, out16, out32, split3); break;
    case 768: hipLaunchKernelGGL((layernorm_kernel<T, 3>), grid, block, 0, s, x, x_stride, gamma, beta, eps, M, out16, out32, split3); break;
    case 1024: hipLaunchKernelGGL((layernorm_kernel<T, 4>), grid, block, 0, s, x, x_stride, gamma, beta, eps, M, out16, out32, split3); break;
    case 1280: hipLaunchKernelGGL((layernorm_kernel<T, 5>), grid, block, 0, s, x, x_stride, gamma, beta, eps, M, out16, out32, split3); break;
    default:
      vidil_set_error("layernorm: D=%d not supported (256/512/768/1024/1280)", D);
      return VIDIL_EUNSUP;
  }
  VIDIL_CHECK_LAUNCH("layernorm");
  return VIDIL_OK;
}

extern "C" int vidil_layernorm(const float* x, int64_t x_stride, const float* gamma, const float* beta, float eps,
                               int32_t M, int32_t D, void* out16, int32_t dtype16, float* out_f32, void* stream) {
  VIDIL_REQUIRE(x && gamma && beta && (out16 || out_f32), "layernorm: null pointer");
  VIDIL_REQUIRE(M > 0, "layernorm: M=%d", M);
  VIDIL_REQUIRE(x_stride % 4 == 0, "layernorm: x_stride must be a multiple of 4");
  if (out16 && dtype16 == VIDIL_DT_FP8)
    return layernorm_launch<fp8>(x, x_stride, gamma, beta, eps, M, D, (fp8*)out16, out_f32, (hipStream_t)stream);
  const int split3 = out16 && (dtype16 & VIDIL_DT_SPLIT3) ? ((dtype16 & VIDIL_DT_SPLIT2) ? 2 : 1) : 0;      // out16 rows are [hi | lo | hi], 3D wide
  VIDIL_REQUIRE(!(dtype16 & VIDIL_DT_SPLIT2) || (dtype16 & VIDIL_DT_SPLIT3), "layernorm: VIDIL_DT_SPLIT2 qualifies VIDIL_DT_SPLIT3");
  VIDIL_DISPATCH_DTYPE(out16 ? (dtype16 & ~(VIDIL_DT_SPLIT3 | VIDIL_DT_SPLIT2)) : VIDIL_DT_F16, "layernorm",
                       return layernorm_launch<T>(x, x_stride, gamma, beta, eps, M, D, (T*)out16, out_f32, (hipStream_t)stream, split3));
}

extern "C" int vidil_split3_f32(const float* x, void* out16, int32_t M, int32_t D, int32_t dtype, void* stream) {
  VIDIL_REQUIRE(x && out16 && M > 0 && D > 0 && D % 8 == 0, "split3: bad args (D %% 8 == 0)");
  const size_t total = (size_t)M * (D / 4);
  VIDIL_DISPATCH_DTYPE(dtype, "split3",
                       hipLaunchKernelGGL(split3_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                                          (T*)out16, M, D));
  VIDIL_CHECK_LAUNCH("split3");
  return VIDIL_OK;
}

extern "C" int vidil_patchify_f32(const float* img, void* out, int32_t B, int32_t S, int32_t ps, int32_t dtype, void* stream) {
  VIDIL_REQUIRE(img && out && B > 0, "patchify_f32: bad args");
  VIDIL_REQUIRE(ps > 0 && S % ps == 0, "patchify_f32: S=%d ps=%d (S%%ps==0 required)", S, ps);
  const int split3 = (dtype & VIDIL_DT_SPLIT3) ? 1 : 0;    // out rows [hi | lo | hi], 3 * round_up(3*ps*ps, 64) wide
  dtype &= ~VIDIL_DT_SPLIT3;
  if (ps % 8 != 0 || split3) {
    const int ldk = (3 * ps * ps + 63) / 64 * 64;
    const size_t n = (size_t)B * (S / ps) * (S / ps) * ldk;
    VIDIL_DISPATCH_DTYPE(dtype, "patchify_f32",
                         hipLaunchKernelGGL((patchify_any_kernel<T, false>), dim3(grid_for(n, 256)), dim3(256), 0,
                                            (hipStream_t)stream, (const void*)img, (T*)out, B, S, ps, ldk, Norm3{}, split3));
    VIDIL_CHECK_LAUNCH("patchify_f32");
    return VIDIL_OK;
  }
  const size_t total = (size_t)B * (S / ps) * (S / ps) * 3 * ps * (ps / 8);
  VIDIL_DISPATCH_DTYPE(dtype, "patchify_f32",
                       hipLaunchKernelGGL(patchify_f32_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                          img, (T*)out, B, S, ps));
  VIDIL_CHECK_LAUNCH("patchify_f32");
  return VIDIL_OK;
}

extern "C" int vidil_patchify_u8(const uint8_t* img, void* out, int32_t B, int32_t S, int32_t ps,
                                 const float* mean3_host, const float* std3_host, int32_t dtype, void* stream) {
  VIDIL_REQUIRE(img && out && mean3_host && std3_host && B > 0, "patchify_u8: bad args");
  VIDIL_REQUIRE(ps > 0 && S % ps == 0, "patchify_u8: S=%d ps=%d (S%%ps==0 required)", S, ps);
  Norm3 nm;
  for (int c = 0; c < 3; ++c) {
    // (x/255 - mean)/std  ==  x * (1/(255*std)) - mean/std ; evaluated in f32 on device
    nm.scale[c] = 1.0f / (255.0f * std3_host[c]);
    nm.shift[c] = -mean3_host[c] / std3_host[c];
  }
  const int split3 = (dtype & VIDIL_DT_SPLIT3) ? 1 : 0;    // out rows [hi | lo | hi], 3 * round_up(3*ps*ps, 64) wide
  dtype &= ~VIDIL_DT_SPLIT3;
  if (ps % 8 != 0 || split3) {
    const int ldk = (3 * ps * ps + 63) / 64 * 64;
    const size_t n = (size_t)B * (S / ps) * (S / ps) * ldk;
    VIDIL_DISPATCH_DTYPE(dtype, "patchify_u8",
                         hipLaunchKernelGGL((patchify_any_kernel<T, true>), dim3(grid_for(n, 256)), dim3(256), 0,
                                            (hipStream_t)stream, (const void*)img, (T*)out, B, S, ps, ldk, nm, split3));
    VIDIL_CHECK_LAUNCH("patchify_u8");
    return VIDIL_OK;
  }
  const size_t total = (size_t)B * (S / ps) * (S / ps) * ps * (ps / 8);
  VIDIL_DISPATCH_DTYPE(dtype, "patchify_u8",
                       hipLaunchKernelGGL(patchify_u8_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                          img, (T*)out, B, S, ps, nm));
  VIDIL_CHECK_LAUNCH("patchify_u8");
  return VIDIL_OK;
}

extern "C" int vidil_set_cls_row(float* x, const float* cls, const float* pos0, int32_t B, int32_t T, int32_t D, void* stream) {
  VIDIL_REQUIRE(x && cls && pos0 && B > 0 && T > 0 && D > 0, "set_cls_row: bad args");
  hipLaunchKernelGGL(set_cls_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, cls, pos0, B, T, D);
  VIDIL_CHECK_LAUNCH("set_cls_row");
  return VIDIL_OK;
}

extern "C" int vidil_embed_tokens(const int32_t* ids, const float* word, const float* pos, float* out, int32_t M,
                                  int32_t T, int32_t pos_off, int32_t D, int32_t vocab, void* stream) {
  VIDIL_REQUIRE(ids && word && pos && out && M > 0 && T > 0 && D % 4 == 0 && vocab > 0, "embed_tokens: bad args");
  hipLaunchKernelGGL(embed_tokens_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, word, pos, out, M, T, pos_off, D, vocab);
  VIDIL_CHECK_LAUNCH("embed_tokens");
  return VIDIL_OK;
}

extern "C" int vidil_gather_rows_f32(const float* x, const int32_t* idx, float* out, int32_t n, int32_t D, void* stream) {
  VIDIL_REQUIRE(x && idx && out && n > 0 && D % 4 == 0, "gather_rows: bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, idx, out, n, D);
  VIDIL_CHECK_LAUNCH("gather_rows");
  return VIDIL_OK;
}

extern "C" int vidil_l2_normalize_rows(float* x, int32_t n, int32_t D, void* stream) {
  VIDIL_REQUIRE(x && n > 0 && D % 4 == 0, "l2_normalize_rows: bad args");
  hipLaunchKernelGGL(l2norm_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, n, D);
  VIDIL_CHECK_LAUNCH("l2_normalize_rows");
  return VIDIL_OK;
}
