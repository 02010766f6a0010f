// developer probe: which CUs does a CU-masked HIP stream run on?  (hipExtStreamCreateWithCUMask; on multi-XCD parts
// the KFD deals the mask bits round robin over the XCDs.)  Build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe
// tools/micro/cu_mask_probe.hip ; run on the GPU box.  Prints, per mask, the number of distinct (xcc, se, cu) slots
// that executed a workgroup and the per-XCD counts.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // keep the workgroup resident for a while so the dispatcher has to spread the grid over every enabled CU
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

static void run(const char* label, const std::vector<unsigned>& mask) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data());
  if (e != hipSuccess) {
    printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", label, hipGetErrorString(e));
    return;
  }
  const int nb = 4096;
  unsigned* d;
  hipMalloc(&d, nb * 8);
  hipMemsetAsync(d, 0xff, nb * 8, s);
  hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(1024), 0, s, d, 200000);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(2 * nb);
  hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::set<unsigned> slots;
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < nb; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    slots.insert(key);
    per_xcc[xcc].insert(key & 0xfff);
  }
  printf("%-28s distinct CUs %3zu | per XCD:", label, slots.size());
  for (auto& kv : per_xcc) printf(" x%u=%zu", kv.first, kv.second.size());
  printf("\n");
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
  auto first_n = [](int n) {
    std::vector<unsigned> m(8, 0u);
    for (int i = 0; i < n; ++i) m[i >> 5] |= 1u << (i & 31);
    return m;
  };
  run("all 256 bits", first_n(256));
  run("first 192 bits", first_n(192));
  run("first 128 bits", first_n(128));
  run("first 64 bits", first_n(64));
  run("first 8 bits", first_n(8));
  {
    std::vector<unsigned> m(8, 0u);
    for (int i = 192; i < 256; ++i) m[i >> 5] |= 1u << (i & 31);
    run("bits 192..255", m);
  }
  {
    std::vector<unsigned> m(8, 0u);
    for (int i = 0; i < 256; ++i)
      if ((i & 7) < 6) m[i >> 5] |= 1u << (i & 31);
    run("bits with (i%8)<6", m);
  }
  return 0;
}
