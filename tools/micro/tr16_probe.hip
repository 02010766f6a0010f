// tr16_probe.hip — what `ds_read_b64_tr_b16` returns (developer probe, gfx950).
// Every lane supplies the address of 4 consecutive 16-bit elements; inside a 16-lane group the 16 addresses are the rows
// (4 lanes each) of a [4][16] block, and lane t of the group receives column t of that block.  The probe gives every
// lane an arbitrary 8-byte-aligned address and checks:  out[l][j] == lds[addr[(l & ~15) + 4*j + ((l & 15) >> 2)] + (l & 3)].
//   hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o tr16_probe.bin && ./tr16_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + addr[l]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int h_addr[64]; short h_out[256];
  for (int l = 0; l < 64; ++l) h_addr[l] = ((l * 37 + 11) % 1000) * 4;
  int* d_addr; short* d_out;
  hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
  hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
  hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want = h_addr[(l & ~15) + 4 * j + ((l & 15) >> 2)] + (l & 3);
      if (h_out[l * 4 + j] != (short)want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, h_out[l * 4 + j], want); ++bad; }
    }
  printf("tr16 probe: %d mismatches of 256\n", bad);
  return bad != 0;
}
