// Developer microbenchmark (round 6): does the ORDER in which one wave per SIMD walks its 16 accumulator tiles change what the
// power-capped socket sustains?  gemm4w issues its 64 MFMAs per K-tile round-robin over 16 tiles (k-step outermost): every MFMA
// reads its C tile from the accumulator file and writes it back.  CHAIN = c consecutive MFMAs on the SAME tile (the k-steps of one
// tile back to back: the hardware forwards the accumulator inside the matrix pipe).  Same flops, same operands, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_chain.bin tools/micro/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAIN>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[4], b[4];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {   // pseudo-random operands: realistic bit toggling, bounded sums
      h = h * 1664525u + 1013904223u; a[k][e] = (__bf16)(((int)(h >> 8) % 2048 - 1024) / 1024.0f);
      h = h * 1664525u + 1013904223u; b[k][e] = (__bf16)(((int)(h >> 8) % 2048 - 1024) / 65536.0f);
    }
  for (int it = 0; it < iters; ++it) {
    // 64 MFMAs = one K-tile of gemm4w: 16 tiles x 4 k-steps
    if constexpr (CHAIN == 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 16; ++t) { acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[(ks + t) & 3], acc[t], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    } else if constexpr (CHAIN == 2) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) { acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kh * 2 + k2], b[(kh * 2 + k2 + t) & 3], acc[t], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[(ks + t) & 3], acc[t], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = __builtin_shufflevector(a[k], a[k], 1, 2, 3, 4, 5, 6, 7, 0);   // operands change every iteration
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}

template <int CHAIN>
void run(int iters) {
  float* d; (void)hipMalloc(&d, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<CHAIN>, dim3(256), dim3(256), 0, 0, d, iters / 4);     // warm (clocks settle under the power limit)
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<CHAIN>, dim3(256), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * 4 * iters * 64 * 2.0 * 32 * 32 * 16;
  printf("one wave per SIMD, 16 accumulator tiles, %d MFMA(s) per tile back to back: %8.2f ms  %7.1f TFLOP/s\n", CHAIN, ms, flop / ms / 1e9);
  (void)hipFree(d);
}

int main() {
  const int iters = 600000;     // ~0.6 s per run at 1.9 PFLOP/s
  for (int rep = 0; rep < 3; ++rep) { run<1>(iters); run<2>(iters); run<4>(iters); }
  return 0;
}
