// Developer microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate with no memory traffic at all.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int e = 0; e < 8; ++e) {   // pseudo-random operands in [-1,1) x [-2^-6, 2^-6): realistic bit toggling, bounded sums
    h = h * 1664525u + 1013904223u; a[e] = (_Float16)(((int)(h >> 8) % 2048 - 1024) / 1024.0f);
    h = h * 1664525u + 1013904223u; b[e] = (_Float16)(((int)(h >> 8) % 2048 - 1024) / 65536.0f);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    a = __builtin_shufflevector(a, a, 1, 2, 3, 4, 5, 6, 7, 0);   // operands change every iteration
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}

template <int NACC>
void run(int waves_per_cu, int iters) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = waves_per_cu * 64 > 512 ? 512 : waves_per_cu * 64;
  const int blocks = 256 * (waves_per_cu * 64 / threads);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * iters * NACC * 2.0 * 32 * 32 * 16;
  printf("waves/CU=%2d  independent acc=%d  %8.3f ms  %7.1f TFLOP/s\n", waves_per_cu, NACC, ms, flop / ms / 1e9);
  hipFree(d);
}

int main() {
  for (int w : {4, 8, 16}) { run<1>(w, 20000); run<2>(w, 20000); run<4>(w, 20000); run<8>(w, 10000); }
  // sustained: a longer run (clocks settle under power limits)
  run<4>(8, 400000);
  return 0;
}
