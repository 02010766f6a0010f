// gemm256w4.hip — 256x256x64 GEMM tile on FOUR waves, one per SIMD, each with the whole 512-entry
// register file (acc 256 + double-buffered fragments 64 + addressing).
//
// Why: with two waves per SIMD (gemm256.hip) the partner's LDS-DMA issue and LDS reads come straight out of
// the matrix pipe's feed and the kernel saturates near 0.85 PFLOP/s.  With one wave per SIMD nothing else
// competes for the SIMD: every k-step the wave issues the 8 ds_read_b128 of the NEXT k-step and a quarter of
// the next K-tile's LDS-DMA, then 16 MFMAs (512 cycles) whose operands were loaded one k-step earlier, so
// LDS latency, DMA issue cost and even the per-K-tile barrier hide behind MFMAs of the same wave.
//
//   * wave (wr, wc) owns the 128x128 quadrant: A half wr x W half wc; 4x4 tiles of
//     v_mfma_f32_32x32x16_f16, operands swapped (D = W_frag * A_frag^T) so a lane holds 4 consecutive
//     output columns;
//   * LDS ring as gemm256.hip: 2 K-tiles x {A0, A1, W0, W1} x 16 KiB, 128-B rows, 16-B slot XOR
//     (row>>1)&7 applied to the DMA source address and to the fragment reads;
//   * per K-tile one `s_waitcnt vmcnt(0) lgkmcnt(0)` + `s_barrier` placed AFTER the reads of the previous
//     tile are in registers and BEFORE the MFMAs that consume its last k-step, i.e. the MFMA stream runs one
//     k-step behind the LDS reads across the barrier;
//   * epilogue: every wave transposes its quadrant through a private 32 KiB of the (idle) ring and stores
//     16 B per lane, 256-512 B contiguous per row.
#include "common.h"

namespace {

constexpr int SLOT = 16384;
constexpr int BUF = 4 * SLOT;
constexpr int LDS_BYTES = 2 * BUF;

__device__ __forceinline__ void glds16(const f16* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int EPI, int ACT>
__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(const vidil_gemm_args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int M = p.M, N = p.N, K = p.K;
  const int lda = p.lda > 0 ? p.lda : K;
  const int tiles_n = (N + 255) >> 8;
  const int tiles_m = (M + 255) >> 8;
  int logical;
  {
    const int nblk = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = logical / tiles_n;
  const int tile_n = logical - tile_m * tiles_n;
  const int m0 = tile_m << 8, n0 = tile_n << 8;
  const int nk = K >> 6;

  // staging sources: half-tile hf of A / W = 1024 16-B chunks = 4 per thread
  int gA[2][4], gW[2][4];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + tid;
      const int r = q >> 3, sl = q & 7;
      const int c = sl ^ ((r >> 1) & 7);
      int ra = m0 + hf * 128 + r;
      ra = ra < M ? ra : M - 1;
      int rw = n0 + hf * 128 + r;
      rw = rw < N ? rw : N - 1;
      gA[hf][i] = ra * lda + c * 8;
      gW[hf][i] = rw * K + c * 8;
    }
  const f16* const baseA = (const f16*)p.A;
  const f16* const baseW = (const f16*)p.W;
  // DMA piece `pc` (0..15) of K-tile `tile`: slot pc>>2 (0 A0, 1 A1, 2 W0, 3 W1), chunk group pc&3
  auto dma_piece = [&](int tile, int pc) {
    if (tile >= nk) return;
    const int slot = pc >> 2, i = pc & 3;
    const f16* base = slot < 2 ? baseA : baseW;
    const int off = slot == 0 ? gA[0][i] : slot == 1 ? gA[1][i] : slot == 2 ? gW[0][i] : gW[1][i];
    glds16(base + tile * 64 + off, smem + (tile & 1) * BUF + slot * SLOT + (i * 256 + wave * 64) * 16);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sw = (l31 >> 1) & 7;
  const int a_off = wr * SLOT + l31 * 128;
  const int w_off = (2 + wc) * SLOT + l31 * 128;

  // fragment double buffer: fa/fw[cur] feed the MFMAs, fa/fw[nxt] receive the next k-step's reads
  f16x8 fa0[4], fw0[4], fa1[4], fw1[4];
  auto read_frags = [&](f16x8(&fa)[4], f16x8(&fw)[4], int tile, int ks) {
    const char* buf = smem + (tile & 1) * BUF;
    const int so = ((ks * 2 + hi) ^ sw) << 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *(const f16x8*)(buf + a_off + i * 4096 + so);
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[j] = *(const f16x8*)(buf + w_off + j * 4096 + so);
  };
  // 16 MFMAs of one k-step; behind each group of 4 (one A row tile) up to two LDS-DMA instructions of K-tile
  // `dtile`, pieces [p0, p0+np), are issued (an LDS-DMA issue costs the wave ~1-2 MFMA slots).
  auto mfma16 = [&](const f16x8(&fa)[4], const f16x8(&fw)[4], int dtile, int p0, int np) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int pc = (np * i) / 4; pc < (np * (i + 1)) / 4; ++pc) dma_piece(dtile, p0 + pc);
    }
  };

  // Schedule (derivation in DESIGN.md §gemm256w4).  K-tile s is DMA'd in three bursts: pieces 0-5 behind the
  // MFMAs of k-step (s-2, 3), 6-11 behind (s-1, 0), 12-15 behind (s-1, 1); it is first read after the barrier
  // that follows k-step (s-1, 2).  That barrier also closes tile s-1's LDS reads (its last fragments were
  // fetched one k-step earlier), so k-step (s-1, 3)'s MFMAs run while tile s's first fragments are in flight
  // and while tile s+1's DMA starts overwriting tile s-1's buffer.
#pragma unroll
  for (int pc = 0; pc < 16; ++pc) dma_piece(0, pc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(fa0, fw0, 0, 0);
#pragma unroll
  for (int pc = 0; pc < 6; ++pc) dma_piece(1, pc);

  for (int t = 0; t < nk; ++t) {
    read_frags(fa1, fw1, t, 1);
    mfma16(fa0, fw0, t + 1, 6, 6);
    read_frags(fa0, fw0, t, 2);
    mfma16(fa1, fw1, t + 1, 12, 4);
    read_frags(fa1, fw1, t, 3);
    mfma16(fa0, fw0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nk) read_frags(fa0, fw0, t + 1, 0);
    mfma16(fa1, fw1, t + 2, 0, 6);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ================================================================================ epilogue
  // acc[it][j][rq*4+e]: row = m_w + it*32 + l31 ; col = n_w + j*32 + rq*8 + hi*4 + e
  const int m_w = m0 + wr * 128;
  const int n_w = n0 + wc * 128;
  if (n_w >= N) return;
  char* ep = smem + wave * (2 * SLOT);  // private 32 KiB transposition buffer of this wave

  if (p.bias != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int col = n_w + j * 32 + rq * 8 + hi * 4;
        if (col + 4 <= N) {
          const f32x4 b4 = *(const f32x4*)(p.bias + col);
#pragma unroll
          for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[it][j][rq * 4 + e] += b4[e];
        }
      }
  }
  auto value = [&](int it, int j, int rq, int e) {
    float v = acc[it][j][rq * 4 + e];
    if constexpr (ACT == VIDIL_ACT_GELU_ERF) v = gelu_erf(v);
    if constexpr (ACT == VIDIL_ACT_QUICK_GELU) v = quick_gelu(v);
    return v;
  };

  int part = 0, head0 = 0;
  if constexpr (EPI == VIDIL_EPI_HEADS) {
    const int hd = p.H * 64;
    part = p.part0 + n_w / hd;          // a 128-column wave tile never straddles a part (H*64 % 128 == 0)
    head0 = (n_w % hd) >> 6;
  }

  if constexpr (EPI == VIDIL_EPI_HEADS) {
    if (part == 2) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = m_w + it * 32 + l31;
        if (m < M) {
          const int b = m / p.T, t = m - b * p.T;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (n_w + j * 32 < N) {
              const int head = head0 + (j >> 1);
              f16* dst = (f16*)p.vt + (((size_t)b * p.H + head) * 64) * (size_t)p.NP + vt_pos(p.t_off + t);
#pragma unroll
              for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  dst[(size_t)((j & 1) * 32 + rq * 8 + hi * 4 + e) * p.NP] = to_f16(value(it, j, rq, e));
            }
          }
        }
      }
      return;
    }
  }

  if constexpr (EPI == VIDIL_EPI_F16 || EPI == VIDIL_EPI_HEADS) {
    // [128 rows][128 cols] halfs, row stride 256 B, 16-B chunk index XOR (row & 15)
    const float scale = (EPI == VIDIL_EPI_HEADS && part == 0) ? p.q_scale : 1.0f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 32 + l31;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f16x4 v = {to_f16(value(it, j, rq, 0) * scale), to_f16(value(it, j, rq, 1) * scale),
                           to_f16(value(it, j, rq, 2) * scale), to_f16(value(it, j, rq, 3) * scale)};
          *(f16x4*)(ep + row * 256 + (((j * 4 + rq) ^ (row & 15)) << 4) + hi * 8) = v;
        }
    }
    const int ch = lane & 15;
#pragma unroll 4
    for (int iter = 0; iter < 32; ++iter) {
      const int row = iter * 4 + (lane >> 4);
      const f16x8 v = *(const f16x8*)(ep + row * 256 + ((ch ^ (row & 15)) << 4));
      const int m = m_w + row;
      const int col = n_w + ch * 8;
      if (m < M && col + 8 <= N) {
        if constexpr (EPI == VIDIL_EPI_F16) {
          *(f16x8*)((f16*)p.out + (size_t)m * p.ldo + col) = v;
        } else {
          const int b = m / p.T, t = m - b * p.T;
          const size_t bh = (size_t)b * p.H + head0 + (ch >> 3);
          if (part == 0) {
            *(f16x8*)((f16*)p.q + (bh * p.Tq_cap + t) * 64 + (ch & 7) * 8) = v;
          } else {
            *(f16x8*)((f16*)p.k + (bh * p.Tk_cap + p.t_off + t) * 64 + (ch & 7) * 8) = v;
          }
        }
      }
    }
  } else {
    // f32: two passes of 64 rows, [64][128] floats, row stride 512 B, 16-B chunk index XOR (row & 7)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int it = pass * 2 + i;
        const int lr = i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const f32x4 v = {value(it, j, rq, 0), value(it, j, rq, 1), value(it, j, rq, 2), value(it, j, rq, 3)};
            *(f32x4*)(ep + lr * 512 + (((j * 8 + rq * 2 + hi) ^ (lr & 7)) << 4)) = v;
          }
      }
      const int ch = lane & 31;
#pragma unroll 4
      for (int iter = 0; iter < 32; ++iter) {
        const int lr = iter * 2 + (lane >> 5);
        f32x4 v = *(const f32x4*)(ep + lr * 512 + ((ch ^ (lr & 7)) << 4));
        const int m = m_w + pass * 64 + lr;
        const int col = n_w + ch * 4;
        if (m < M && col + 4 <= N) {
          if constexpr (EPI == VIDIL_EPI_F32) {
            const size_t o = (size_t)m * p.ldo + col;
            if (p.resid != nullptr) v += *(const f32x4*)(p.resid + o);
            *(f32x4*)((float*)p.out + o) = v;
          } else {  // EPI_PATCH
            const int b = m / p.tpi, t = m - b * p.tpi;
            v += *(const f32x4*)(p.pos + (size_t)(t + 1) * N + col);
            *(f32x4*)((float*)p.out + ((size_t)m + b + 1) * p.ldo + col) = v;
          }
        }
      }
    }
  }
}

template <int EPI, int ACT>
int launch_w4(const vidil_gemm_args& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm256w4_kernel<EPI, ACT>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      vidil_set_error("gemm256w4: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return VIDIL_ELAUNCH;
    }
    attr_set = true;
  }
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), LDS_BYTES, s, a);
  VIDIL_CHECK_LAUNCH("gemm256w4");
  return VIDIL_OK;
}

}  // namespace

int vidil_gemm256w4_launch(const vidil_gemm_args& a, hipStream_t s) {
  switch (a.epi) {
    case VIDIL_EPI_F16:
      if (a.act == VIDIL_ACT_NONE) return launch_w4<VIDIL_EPI_F16, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch_w4<VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF>(a, s);
      return launch_w4<VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_F32:
      if (a.act == VIDIL_ACT_NONE) return launch_w4<VIDIL_EPI_F32, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch_w4<VIDIL_EPI_F32, VIDIL_ACT_GELU_ERF>(a, s);
      return launch_w4<VIDIL_EPI_F32, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_HEADS:
      return launch_w4<VIDIL_EPI_HEADS, VIDIL_ACT_NONE>(a, s);
    default:
      return launch_w4<VIDIL_EPI_PATCH, VIDIL_ACT_NONE>(a, s);
  }
}
