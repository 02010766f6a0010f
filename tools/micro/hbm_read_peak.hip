// hbm_read_peak.hip — what a read-only stream reaches on this part (developer probe, gfx950): 2.2 GB read once by
//   (a) persistent workgroups, UNR x 16-byte global loads per lane in flight, grid-stride over 1-KiB wave chunks;
//   (b) the same traffic as LDS-DMA (global_load_lds_dwordx4) into a ring, vmcnt-counted;
//   (c) one-shot workgroups of 256 threads that read 50 KiB each (the decode cross-attention's unit) and exit.
// Prints TB/s per variant.   hipcc --offload-arch=gfx950 -O3 hbm_read_peak.hip -o hbm_read_peak.bin && ./hbm_read_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNR>
__global__ __launch_bounds__(256) void k_loads(const f32x4* __restrict__ src, size_t n16, float* out) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
    f32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u][0] + v[u][3];
  }
  if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_dma(const char* __restrict__ src, size_t bytes, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // every wave streams its own 1-KiB chunks into a private ring of 16 KiB (16 pieces in flight)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t nchunk = bytes >> 10;
  const size_t stride = (size_t)gridDim.x * 4;
  size_t c = (size_t)blockIdx.x * 4 + wave;
  char* ring = smem + wave * 16384;
  int slot = 0;
  for (; c < nchunk; c += stride) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (c << 10) + lane * 16),
                                     (__attribute__((address_space(3))) void*)(ring + slot * 1024), 16, 0, 0);
    slot = (slot + 1) & 15;
    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ring[lane] == 123 && src == nullptr) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_units(const f32x4* __restrict__ src, float* out) {
  // one unit = 50 KiB = 3200 x 16 B: 256 threads x 12.5 loads, all issued before any use
  const f32x4* p = src + (size_t)blockIdx.x * 3200;
  f32x4 v[13];
#pragma unroll
  for (int u = 0; u < 13; ++u) {
    const int i = threadIdx.x + u * 256;
    v[u] = p[i < 3200 ? i : threadIdx.x];
  }
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 13; ++u) acc += v[u][0] + v[u][3];
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)43008 * 3200 * 16;     // 2.2 GB
  char* src; float* out;
  hipMalloc(&src, bytes); hipMalloc(&out, 64);
  hipMemset(src, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, ms * 100.f, bytes / (ms * 1e-4) / 1e12);
  };
  const size_t n16 = bytes / 16;
  for (int g : {512, 1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, 64, "loads x4 in flight, grid %d", g);
    time(nm, [&] { hipLaunchKernelGGL(k_loads<4>, dim3(g), dim3(256), 0, 0, (const f32x4*)src, n16, out); });
    snprintf(nm, 64, "loads x8 in flight, grid %d", g);
    time(nm, [&] { hipLaunchKernelGGL(k_loads<8>, dim3(g), dim3(256), 0, 0, (const f32x4*)src, n16, out); });
  }
  hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int g : {256, 512}) {
    char nm[64];
    snprintf(nm, 64, "LDS-DMA ring 16 x 1 KiB per wave, grid %d", g);
    time(nm, [&] { hipLaunchKernelGGL(k_dma, dim3(g), dim3(256), 65536, 0, (const char*)src, bytes, out); });
  }
  time("one-shot 50-KiB units (43,008 workgroups)", [&] { hipLaunchKernelGGL(k_units, dim3(43008), dim3(256), 0, 0, (const f32x4*)src, out); });
  return 0;
}
