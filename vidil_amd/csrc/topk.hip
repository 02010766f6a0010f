// topk.hip — sorted top-k of every row of an f32 matrix (k <= 128, rows of up to 38,400 values).
//
// Used by the BLIP retrieval backend of the visual tokenizer (run_visual_tokenization.py:277-293:
// `sims.topk(k=config['k_test'])` per frame, then an ITM re-rank of those k texts).  One workgroup per row: the row
// sits in LDS, the k winners are extracted by repeated block-wide arg-max, ordered by (value descending, index
// ascending) — torch.topk leaves the order of exact ties unspecified; this one is deterministic.
#include "common.h"

namespace {

struct Best {
  float v;
  int i;
};
__device__ __forceinline__ Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, int64_t row_stride, int N, int k,
                                                        float* __restrict__ out_v, int32_t* __restrict__ out_i) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* row = (float*)smem;
  __shared__ Best wbest[4];
  const int tid = threadIdx.x;
  const float* xr = x + (size_t)blockIdx.x * row_stride;
  for (int i = tid; i < N; i += 256) row[i] = xr[i];
  __syncthreads();
  for (int it = 0; it < k; ++it) {
    Best me{-INFINITY, 0x7fffffff};
    for (int i = tid; i < N; i += 256) me = better(me, Best{row[i], i});
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) me = better(me, Best{__shfl_xor(me.v, o, 64), __shfl_xor(me.i, o, 64)});
    if ((tid & 63) == 0) wbest[tid >> 6] = me;
    __syncthreads();
    if (tid == 0) {
      const Best w = better(better(wbest[0], wbest[1]), better(wbest[2], wbest[3]));
      out_v[(size_t)blockIdx.x * k + it] = w.v;
      out_i[(size_t)blockIdx.x * k + it] = w.i == 0x7fffffff ? -1 : w.i;
      if (w.i != 0x7fffffff) row[w.i] = -INFINITY;   // (a row of fewer than k finite values yields -inf / -1 tails)
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int vidil_topk_rows(const float* x, int64_t row_stride, int32_t R, int32_t N, int32_t k, float* out_v,
                               int32_t* out_i, void* stream) {
  VIDIL_REQUIRE(x && out_v && out_i, "topk_rows: null pointer");
  VIDIL_REQUIRE(R > 0 && N > 0 && k > 0 && k <= 128 && k <= N && row_stride >= N, "topk_rows: R=%d N=%d k=%d stride=%ld", R, N, k,
                (long)row_stride);
  const size_t lds = (size_t)N * 4;
  if (lds > 150 * 1024) {
    vidil_set_error("topk_rows: rows of %d values do not fit the LDS row buffer (<= 38400)", N);
    return VIDIL_EUNSUP;
  }
  static std::atomic<unsigned long long> attr_set{0};   // (one bit per device that has the opt-in: vidil_lds_opt_in)
  if (const int rc_ = vidil_lds_opt_in(attr_set, (const void*)topk_rows_kernel, 150 * 1024, "topk_rows")) return rc_;
  hipLaunchKernelGGL(topk_rows_kernel, dim3(R), dim3(256), lds, (hipStream_t)stream, x, row_stride, N, k, out_v, out_i);
  VIDIL_CHECK_LAUNCH("topk_rows");
  return VIDIL_OK;
}
