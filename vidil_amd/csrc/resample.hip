// resample.hip — Pillow's antialiased bicubic resize of 8-bit RGB frames, one separable pass per launch.
//
// Replaces: transforms.Resize((S,S), BICUBIC) on a PIL image
// (run_video_CapFilt.py:128-134) and HF CLIPProcessor's shortest-edge resize + centre crop
// (run_visual_tokenization.py:138-142), both of which are PIL Image.resize(..., BICUBIC) = ImagingResample
// (Pillow src/libImaging/Resample.c): horizontal pass over the needed source rows into a u8 intermediate,
// then the vertical pass, each output byte
//     clip8((2^21 + sum_i pixel_i * k_i) >> 22),   k_i = 22-bit fixed-point filter weights
// The weights come from the host (double precision set-up restated in vidil_amd/preprocess.py); the device
// part is integer arithmetic, so the result is BIT-EXACT with Pillow.  HBM-bound byte work: one thread per
// output byte, vertical-pass accesses are fully coalesced, horizontal-pass windows overlap in L1/L2.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct ResampleP {
  const uint8_t* src;
  uint8_t* dst;
  const int32_t* bounds;   // [n_out][2] = (first tap, tap count)
  const int32_t* coeffs;   // [n_out][ksize]
  int B, in_h, in_w, out_h, out_w, ksize, src_row0;
};

__device__ __forceinline__ uint8_t clip8(int acc) {
  const int v = acc >> PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[b][y][xx][c] = sum_x src[b][src_row0+y][lo(xx)+x][c] * k[xx][x]
__global__ __launch_bounds__(256) void resample_h_kernel(const ResampleP p) {
  const size_t row_bytes = (size_t)p.out_w * 3;
  const size_t total = (size_t)p.B * p.out_h * row_bytes;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % row_bytes);
    const size_t by = i / row_bytes;
    const int y = (int)(by % p.out_h);
    const int b = (int)(by / p.out_h);
    const int xx = j / 3, c = j - xx * 3;
    const int lo = p.bounds[2 * xx], cnt = p.bounds[2 * xx + 1];
    const int32_t* __restrict__ k = p.coeffs + (size_t)xx * p.ksize;
    const uint8_t* __restrict__ s = p.src + (((size_t)b * p.in_h + p.src_row0 + y) * p.in_w + lo) * 3 + c;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < cnt; ++x) acc += (int)s[x * 3] * k[x];
    p.dst[i] = clip8(acc);
  }
}

// Same pass, LDS-tiled: a workgroup stages HR consecutive source rows (coalesced 4-byte loads; the rows of one
// frame are contiguous) and the whole weight table once, then every output byte takes its taps from LDS.  The
// generic kernel above issues one scattered byte load per tap (13 taps per output byte at 360x640 -> 224^2).
constexpr int HR = 4;
__global__ __launch_bounds__(256) void resample_h_lds_kernel(const ResampleP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int in_row = p.in_w * 3;
  const int in_row4 = (in_row + 3) & ~3;
  uint8_t* rows = (uint8_t*)smem;                                   // [HR][in_row4]
  int32_t* kt = (int32_t*)(smem + HR * in_row4);                    // [out_w][ksize]
  int32_t* bt = kt + p.out_w * p.ksize;                             // [out_w][2]
  const int tid = threadIdx.x;
  const int nrows_total = p.B * p.out_h;
  const int r0 = blockIdx.x * HR;
  for (int i = tid; i < p.out_w * p.ksize; i += 256) kt[i] = p.coeffs[i];
  for (int i = tid; i < p.out_w * 2; i += 256) bt[i] = p.bounds[i];
  for (int r = 0; r < HR; ++r) {
    const int gr = r0 + r;
    if (gr >= nrows_total) break;
    const int b = gr / p.out_h, y = gr - b * p.out_h;
    const uint8_t* s = p.src + ((size_t)b * p.in_h + p.src_row0 + y) * in_row;
    // byte-wise head / tail around the 4-byte aligned middle of the row
    const int mis = (int)((4 - ((uintptr_t)s & 3)) & 3);
    const int head = mis < in_row ? mis : in_row;
    if (tid < head) rows[r * in_row4 + tid] = s[tid];
    const int nwords = (in_row - head) >> 2;
    for (int w = tid; w < nwords; w += 256) {
      const uint32_t v = *(const uint32_t*)(s + head + 4 * w);
      uint8_t* d = rows + r * in_row4 + head + 4 * w;
      d[0] = (uint8_t)v; d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)(v >> 16); d[3] = (uint8_t)(v >> 24);
    }
    const int done = head + 4 * nwords;
    if (tid < in_row - done) rows[r * in_row4 + done + tid] = s[done + tid];
  }
  __syncthreads();
  const int out_row = p.out_w * 3;
  for (int i = tid; i < HR * out_row; i += 256) {
    const int r = i / out_row, j = i - r * out_row;
    const int gr = r0 + r;
    if (gr >= nrows_total) break;
    const int xx = j / 3, c = j - xx * 3;
    const int lo = bt[2 * xx], cnt = bt[2 * xx + 1];
    const int32_t* k = kt + xx * p.ksize;
    const uint8_t* s = rows + r * in_row4 + lo * 3 + c;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < cnt; ++x) acc += (int)s[x * 3] * k[x];
    p.dst[(size_t)gr * out_row + j] = clip8(acc);
  }
}

// dst[b][yy][j] = sum_y src[b][lo(yy)+y][j] * k[yy][y]      (j = byte within the row: in_w == out_w)
__global__ __launch_bounds__(256) void resample_v_kernel(const ResampleP p) {
  const size_t row_bytes = (size_t)p.out_w * 3;
  const size_t total = (size_t)p.B * p.out_h * row_bytes;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % row_bytes);
    const size_t byy = i / row_bytes;
    const int yy = (int)(byy % p.out_h);
    const int b = (int)(byy / p.out_h);
    const int lo = p.bounds[2 * yy], cnt = p.bounds[2 * yy + 1];
    const int32_t* __restrict__ k = p.coeffs + (size_t)yy * p.ksize;
    const uint8_t* __restrict__ s = p.src + ((size_t)b * p.in_h + lo) * row_bytes + j;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) acc += (int)s[(size_t)y * row_bytes] * k[y];
    p.dst[i] = clip8(acc);
  }
}

// Vertical pass, four output bytes per thread (rows of a multiple of 4 bytes, 4-byte aligned buffers): 32-bit
// loads and stores, 256 B per wave-instruction instead of 64.
__global__ __launch_bounds__(256) void resample_v4_kernel(const ResampleP p) {
  const size_t row_words = (size_t)p.out_w * 3 / 4;
  const size_t total = (size_t)p.B * p.out_h * row_words;
  const uint32_t* __restrict__ src = (const uint32_t*)p.src;
  uint32_t* __restrict__ dst = (uint32_t*)p.dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % row_words);
    const size_t byy = i / row_words;
    const int yy = (int)(byy % p.out_h);
    const int b = (int)(byy / p.out_h);
    const int lo = p.bounds[2 * yy], cnt = p.bounds[2 * yy + 1];
    const int32_t* __restrict__ k = p.coeffs + (size_t)yy * p.ksize;
    const uint32_t* __restrict__ s = src + ((size_t)b * p.in_h + lo) * row_words + j;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
    for (int y = 0; y < cnt; ++y) {
      const uint32_t v = s[(size_t)y * row_words];
      const int kk = k[y];
      a0 += (int)(v & 0xff) * kk;
      a1 += (int)((v >> 8) & 0xff) * kk;
      a2 += (int)((v >> 16) & 0xff) * kk;
      a3 += (int)(v >> 24) * kk;
    }
    dst[i] = (uint32_t)clip8(a0) | ((uint32_t)clip8(a1) << 8) | ((uint32_t)clip8(a2) << 16) | ((uint32_t)clip8(a3) << 24);
  }
}

}  // namespace

extern "C" int vidil_resample_u8(const uint8_t* src, uint8_t* dst, int32_t B, int32_t in_h, int32_t in_w, int32_t out_h,
                                 int32_t out_w, int32_t vertical, const int32_t* bounds, const int32_t* coeffs,
                                 int32_t ksize, int32_t src_row0, void* stream) {
  VIDIL_REQUIRE(src && dst && bounds && coeffs, "resample_u8: null pointer");
  VIDIL_REQUIRE(B > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && ksize > 0, "resample_u8: bad shape");
  if (vertical) {
    VIDIL_REQUIRE(in_w == out_w && src_row0 == 0, "resample_u8: the vertical pass keeps the width (in_w=%d out_w=%d)", in_w, out_w);
  } else {
    VIDIL_REQUIRE(src_row0 >= 0 && src_row0 + out_h <= in_h, "resample_u8: rows %d..%d outside the %d source rows", src_row0,
                  src_row0 + out_h - 1, in_h);
  }
  const ResampleP p{src, dst, bounds, coeffs, B, in_h, in_w, out_h, out_w, ksize, src_row0};
  const size_t total = (size_t)B * out_h * out_w * 3;
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;   // grid-stride beyond 64 blocks per CU
  if (vertical) {
    if ((out_w * 3) % 4 == 0 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 3) == 0) {
      size_t b4 = (total / 4 + 255) / 256;
      if (b4 > 256 * 64) b4 = 256 * 64;
      hipLaunchKernelGGL(resample_v4_kernel, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, p);
    } else {
      hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    }
  } else {
    const size_t lds = (size_t)HR * ((in_w * 3 + 3) & ~3) + (size_t)out_w * ksize * 4 + (size_t)out_w * 8;
    if (lds <= 64 * 1024) {
      const int nrows = B * out_h;
      hipLaunchKernelGGL(resample_h_lds_kernel, dim3((unsigned)((nrows + HR - 1) / HR)), dim3(256), lds, (hipStream_t)stream, p);
    } else {
      hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    }
  }
  VIDIL_CHECK_LAUNCH("resample_u8");
  return VIDIL_OK;
}
