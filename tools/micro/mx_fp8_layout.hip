// Developer probe (GPU box): operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) A and B.
// Hypothesis H0: lane l holds row (l % 32), k = 32 * (l / 32) + byte index (32 contiguous k per lane).
// Hypothesis H1: 16-byte halves interleave: bytes 0-15 -> k = 16 * (l / 32) + b, bytes 16-31 -> k = 32 + 16 * (l / 32) + (b - 16).
// Fills A[32][64], B[64][32] with small integers (exact in e4m3), runs one MFMA per hypothesis, compares with the
// integer reference.  Prints which hypothesis matches.    hipcc --offload-arch=gfx950 -O2 mx_fp8_layout.hip -o mx_fp8_layout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __host__ inline uint8_t to_e4m3(int v) {   // small integers 0..8 and their negatives, exact
  // e4m3fn: sign(1) exp(4, bias 7) mant(3)
  if (v == 0) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  int a = v < 0 ? -v : v;
  int e = 0;
  while ((a >> (e + 1)) != 0) ++e;                 // floor(log2 a)
  int mant = ((a << 3) >> e) & 7;                  // a = 2^e * (1 + mant/8) for a < 16 with <= 3 fractional bits
  return s | (uint8_t)((e + 7) << 3) | (uint8_t)mant;
}

__global__ void probe(const uint8_t* A /*[32][64]*/, const uint8_t* B /*[64][32] stored as Bt[32][64]*/, float* out /*[2][32][32]*/) {
  const int lane = threadIdx.x;
  const int r = lane & 31, hi = lane >> 5;
  for (int hyp = 0; hyp < 2; ++hyp) {
    uint8_t a[32], b[32];
    for (int j = 0; j < 32; ++j) {
      int k = hyp == 0 ? 32 * hi + j : (j < 16 ? 16 * hi + j : 32 + 16 * hi + (j - 16));
      a[j] = A[r * 64 + k];
      b[j] = B[r * 64 + k];
    }
    v8i av, bv;
    for (int w = 0; w < 8; ++w) {
      av[w] = a[4 * w] | (a[4 * w + 1] << 8) | (a[4 * w + 2] << 16) | (a[4 * w + 3] << 24);
      bv[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
    }
    v16f c = {0};
    // cbsz = 0 (A fp8 e4m3), blgp = 0 (B fp8 e4m3), scales 2^0 (E8M0 127)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    // C layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * hi
    for (int reg = 0; reg < 16; ++reg) out[(hyp * 32 + ((reg & 3) + 8 * (reg >> 2) + 4 * hi)) * 32 + r] = c[reg];
  }
}

int main() {
  uint8_t hA[32 * 64], hB[32 * 64];
  int iA[32 * 64], iB[32 * 64];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) {
    iA[i] = rand() % 9 - 4; iB[i] = rand() % 9 - 4;
    hA[i] = to_e4m3(iA[i]); hB[i] = to_e4m3(iB[i]);
  }
  uint8_t *dA, *dB; float* dO;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dO, 2 * 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO);
  float hO[2 * 32 * 32];
  hipMemcpy(hO, dO, sizeof(hO), hipMemcpyDeviceToHost);
  // which operand is "rows" of the result? D[i][j] = sum_k Aop[i][k] * Bop[j][k]; the first MFMA operand supplies i (C rows).
  for (int hyp = 0; hyp < 2; ++hyp) {
    int bad_ab = 0, bad_ba = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        int ref_ab = 0, ref_ba = 0;
        for (int k = 0; k < 64; ++k) { ref_ab += iA[i * 64 + k] * iB[j * 64 + k]; ref_ba += iB[i * 64 + k] * iA[j * 64 + k]; }
        bad_ab += hO[(hyp * 32 + i) * 32 + j] != (float)ref_ab;
        bad_ba += hO[(hyp * 32 + i) * 32 + j] != (float)ref_ba;
      }
    printf("hypothesis H%d: mismatches with rows<-first operand %d, rows<-second operand %d (of 1024)\n", hyp, bad_ab, bad_ba);
  }
  return 0;
}
