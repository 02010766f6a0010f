// gemm128x256.hip — the large-M GEMM again, restructured so that TWO independent workgroups share a CU.
//
// Why: gemm256.hip (one 8-wave workgroup per CU, all registers and 128 KiB of LDS) keeps every wave of the CU in
// the same phase.  Measured per 256x256 tile at K = 768: 24.6k cycles of MFMA issue, ~12k cycles parked at the 25
// s_waitcnt / s_barrier points of the main loop (both waves of a SIMD stop at the same barrier, so nothing feeds the
// matrix pipe while fragments are re-read), and then 8k (16-bit rows) to 29k (f32 residual read-modify-write,
// bounded by the CU's ~17 B/clk memory path) cycles of epilogue during which the pipe idles as well.
//
// Here a workgroup is 4 waves (one per SIMD, 256 VGPRs each) computing a 128 x 256 tile; two such workgroups are
// resident per CU and nothing synchronises them, so one workgroup's barrier waits and its whole epilogue (HBM
// traffic, GELU VALU work) overlap the other's MFMA stream.
//   * A (128 rows x 64 k per K-tile, 16 KiB) goes through a 4-deep LDS ring filled by global_load_lds_dwordx4, three
//     K-tiles ahead, ONE barrier per K-tile (4 waves);
//   * W never touches LDS: the host stores it in FRAGMENT TILES (vidil_gemm_args.W_tiled: per 64 columns x K-tile,
//     the 8 KiB a wave needs in operand order) and every wave loads its fragments straight from L2 into registers
//     with contiguous 1-KiB wave loads, one K-tile ahead, refilling each k-step's registers right after the k-step's
//     MFMAs.  That halves the LDS bytes per MFMA of gemm256 (A only: 16 fragment reads per 32 MFMAs instead of 24)
//     and removes half of the LDS-DMA traffic;
//   * the W loads are inline asm (hipcc drains vmcnt(0) before any use of an ordinary load while LDS-DMA is in
//     flight) and are counted by hand together with the LDS-DMA of A: per k-step the wave issues its W loads, then its
//     DMA pieces, always in that order, so `s_waitcnt vmcnt(10)` (16-bit: 2 + 1 per k-step; fp8: vmcnt(8), 4 + 2) in
//     front of a k-step guarantees that step's W registers and — being older — every A tile up to the current one;
//   * wave tile, accumulator layout and epilogue are gemm256's (gemm_epilogue.h): same k order, same roundings ->
//     bit-identical results.
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int RING = 4;            // K-tiles of A resident in LDS (three in flight ahead of the one being multiplied)
constexpr int ASLOT = 16384;       // one A K-tile: 128 rows x 128 B
constexpr int LDS_BYTES = RING * ASLOT;

__device__ __forceinline__ void glds16(const void* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
// 16 bytes per lane straight into registers, invisible to hipcc's waitcnt bookkeeping (counted by hand, see above)
__device__ __forceinline__ void gload16(i32x4& dst, const char* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T> struct WFrag;    // a W fragment from the raw 16-byte loads
template <typename T> struct WFrag {
  static __device__ __forceinline__ typename Mma<T>::Frag get(const i32x4 (&raw)[8], int j, int ks) {
    return __builtin_bit_cast(typename Mma<T>::Frag, raw[j * 4 + ks]);
  }
};
template <> struct WFrag<fp8> {
  static __device__ __forceinline__ i32x8 get(const i32x4 (&raw)[8], int j, int ks) {
    return __builtin_shufflevector(raw[(j * 2 + ks) * 2], raw[(j * 2 + ks) * 2 + 1], 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

template <typename T, typename TO, int EPI, int ACT, bool FOLD, bool STATS>
__global__ __launch_bounds__(256, 2) void gemm128x256_kernel(const vidil_gemm_args p) {
  using f16 = TO;                       // (the epilogue text is written in terms of "the 16-bit output type")
  using f16x4 = typename Elt<TO>::x4;
  using f16x8 = typename Elt<TO>::x8;
  using Frag = typename Mma<T>::Frag;
  constexpr int KS = Mma<T>::KS;
  constexpr int ESZ = sizeof(T);
  constexpr int KT = 128 / ESZ;
  constexpr int WPK = 8 / KS;            // raw W loads per k-step
  constexpr int DPK = 4 / KS;            // A DMA pieces per k-step
  constexpr int VMW = (12 * KS - 8) / KS;   // ops issued after a k-step's W loads until that k-step comes round again
  static_assert(!FOLD || ESZ == 2, "the LayerNorm fold is a 16-bit feature");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // column block of this wave (64 output columns)
  const int hi = lane >> 5, l31 = lane & 31;

  const int M = p.M, N = p.N, K = p.K;
  const int lda = p.lda > 0 ? p.lda : K;
  const int tiles_n = (N + 255) >> 8;
  const int tiles_m = (M + 127) >> 7;
  const int ncb = (N + 63) >> 6;
  const int nk = K / KT;
  // persistent workgroups, XCD-ranged tile walk (as gemm256): workgroup b lives on XCD b % 8
  int logical, remaining;
  const int tile_step = gridDim.x >= 8 ? (gridDim.x >> 3) : 1;
  {
    const int nblk = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7, slot = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    remaining = (xcd < r ? q + 1 : q) - slot;
  }
  if (remaining <= 0) return;

  int gA[4];        // this thread's four 16-byte chunks of an A K-tile (element offsets from p.A)
  const char* wp;   // this lane's position in the wave's first W fragment tile of the current output tile
  int m0, n0;
  auto setup_tile = [&](int lt) {
    const int tile_m = lt / tiles_n;
    const int tile_n = lt - tile_m * tiles_n;
    m0 = tile_m << 7;
    n0 = tile_n << 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + tid;
      const int r = q >> 3, sl = q & 7;
      const int c = sl ^ ((r >> 1) & 7);
      int ra = m0 + r;
      ra = ra < M ? ra : M - 1;
      gA[i] = ra * lda + c * (16 / ESZ);
    }
    int cb = tile_n * 4 + wave;
    cb = cb < ncb ? cb : ncb - 1;       // (column blocks past N: the wave computes a duplicate and stores nothing)
    wp = (const char*)p.W_tiled + ((size_t)cb * nk) * 8192 + lane * 16;
  };
  setup_tile(logical);
  const T* const baseA = (const T*)p.A;
  auto issue_a = [&](int tile, int piece) {     // piece 0..3 of A K-tile `tile` (clamped: counts stay uniform)
    const int tt = tile < nk ? tile : nk - 1;
    glds16(baseA + tt * KT + gA[piece], smem + (tile & (RING - 1)) * ASLOT + piece * 4096 + wave * 1024);
  };
  i32x4 wraw[8];
  auto issue_w = [&](int tile, int first, int count) {   // raw loads [first, first+count) of W K-tile `tile`
    const int tt = tile < nk ? tile : nk - 1;
    const char* src = wp + (size_t)tt * 8192;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r >= first && r < first + count) gload16(wraw[r], src + r * 1024);
  };
  // raw indices a k-step's MFMAs read (and which are refilled behind them): 16-bit {ks, 4+ks}; fp8 {4j + 2ks, +1}
  auto refill_w = [&](int tile, int ks) {
    const int tt = tile < nk ? tile : nk - 1;
    const char* src = wp + (size_t)tt * 8192;
    if constexpr (KS == 4) {
      gload16(wraw[ks], src + ks * 1024);
      gload16(wraw[4 + ks], src + (4 + ks) * 1024);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        gload16(wraw[(j * 2 + ks) * 2], src + ((j * 2 + ks) * 2) * 1024);
        gload16(wraw[(j * 2 + ks) * 2 + 1], src + ((j * 2 + ks) * 2 + 1) * 1024);
      }
    }
  };

  f32x16 acc[4][2];
  float st_s[4], st_ss[4];
  f32x2 st_raw[4][2];   // FOLD: this lane's share of the producer's row partials (loaded at the tile top)
  const int sw = (l31 >> 1) & 7;
  const int a_off = l31 * 128;

  // ---- prologue of a tile: A(0), A(1) go out before the previous tile's epilogue (LDS-DMA needs no registers);
  //      W(0) — 32 registers that the epilogue needs — and A(2) (whose slot is the epilogue's scratch) follow at the
  //      top of the K loop
  auto prologue = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a(1, i);
  };
  prologue();

  for (;;) {   // ======================================================================== one output tile
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[it][j][r] = 0.f;
    if constexpr (FOLD) {
      const int nparts = K >> 6;            // <= 16
      const f32x2* stats_in = (const f32x2*)p.ln_stats;
      const int part0 = wave + 4 * hi;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        int row = m0 + it * 32 + l31;
        row = row < M ? row : M - 1;
        st_raw[it][0] = part0 < nparts ? stats_in[(size_t)row * nparts + part0] : f32x2{0.f, 0.f};
        st_raw[it][1] = part0 + 8 < nparts ? stats_in[(size_t)row * nparts + part0 + 8] : f32x2{0.f, 0.f};
      }
    }
    issue_w(0, 0, 8);
    // everything issued so far (the prologue, the previous epilogue's traffic, W(0)) has landed after this; from here
    // on only the uniform per-k-step pattern is in flight
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is out of the previous epilogue: its LDS scratch (slots 2, 3) is free
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a(2, i);

    for (int t = 0; t < nk; ++t) {
      const char* abuf = smem + (t & (RING - 1)) * ASLOT + a_off;
      // A(t): this wave's pieces are older than the newest VMW operations -> landed; the barrier publishes every
      // wave's pieces and orders the reads of tile t-1 (done: lgkmcnt) before the DMA below reuses its slot
      wait_vm<VMW>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // A fragments: one k-step ahead inside the K-tile for 16-bit operands (the registers allow it); fp8 fragments
      // are twice as large and are read right before their MFMAs (the other workgroup's wave covers the latency)
      constexpr int NB = ESZ == 2 ? 2 : 1;
      Frag a[NB][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[0][i] = Mma<T>::load(abuf + i * 4096, 0, hi, sw);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks > 0) wait_vm<VMW>();      // this k-step's W registers (ks == 0: the wait above)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NB == 2) {
          if (ks + 1 < KS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[(ks + 1) & 1][i] = Mma<T>::load(abuf + i * 4096, ks + 1, hi, sw);
          }
          __builtin_amdgcn_sched_barrier(0);   // the look-ahead reads go out BEFORE this k-step's MFMAs (hipcc sinks them otherwise)
        } else if (ks > 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) a[0][i] = Mma<T>::load(abuf + i * 4096, ks, hi, sw);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = Mma<T>::mma(WFrag<T>::get(wraw, j, ks), a[ks & (NB - 1)][i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        refill_w(t + 1, ks);
#pragma unroll
        for (int d = 0; d < DPK; ++d) issue_a(t + 3, ks * DPK + d);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // The last K-tile's refills are never read, so hipcc considers wraw dead from the moment the asm loads were
    // issued and may hand those registers to other values while the loads are still in flight (seen as ONE wrong tile
    // in 5,544): keep them live until the drain above has completed.
#pragma unroll
    for (int r = 0; r < 8; ++r) asm volatile("" ::"v"(wraw[r]));
    __builtin_amdgcn_s_barrier();

    if constexpr (FOLD) {
      f32x2* stats = (f32x2*)smem;      // ring slot 0 (idle: nothing in flight, the prologue below comes after)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float s0 = st_raw[it][0][0] + st_raw[it][1][0], ss0 = st_raw[it][0][1] + st_raw[it][1][1];
        const float s = s0 + __shfl_xor(s0, 32, 64);
        const float ss = ss0 + __shfl_xor(ss0, 32, 64);
        if (hi == 0) stats[wave * 128 + it * 32 + l31] = f32x2{s, ss};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const float inv_k = 1.0f / (float)K;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const f32x2 v = stats[w * 128 + it * 32 + l31];
          s += v[0];
          ss += v[1];
        }
        const float mean = s * inv_k;
        float var = ss * inv_k - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rstd = 1.0f / sqrtf(var + p.ln_eps);
        st_s[it] = rstd;
        st_ss[it] = mean * rstd;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }

    const int m_w = m0;
    const int n_w = n0 + wave * 64;
    const bool more = EPI != VIDIL_EPI_HEADS && remaining > tile_step;   // uniform (the per-head scatter: one tile per workgroup,
                                                                        // as in gemm256 — its epilogue has no registers to spare)
    if (more) {
      logical += tile_step;
      remaining -= tile_step;
      setup_tile(logical);
      prologue();                        // slots 0 and 1; the epilogue transposes through slots 2 and 3
    }
    do {
      char* const ep = smem + 2 * ASLOT + wave * 8192;   // this wave's 8-KiB transposition scratch (ring slots 2, 3)
constexpr bool RLN = false;   // (residual LayerNorm: 256x256 kernel only)
#include "gemm_epilogue.inc"
    } while (0);
    if (!more) break;
  }
}

template <typename T, int EPI, int ACT, bool FOLD = false, typename TO = T, bool STATS = false>
int launch_w4(const vidil_gemm_args& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm128x256_kernel<T, TO, EPI, ACT, FOLD, STATS>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      vidil_set_error("gemm128x256: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return VIDIL_ELAUNCH;
    }
    attr_set = true;
  }
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
      n = 256;
    num_cu = n & ~7;
  }
  const int ntiles = ((a.M + 127) / 128) * ((a.N + 255) / 256);
  const int cap = 2 * num_cu;     // two workgroups per CU
  const int grid = EPI == VIDIL_EPI_HEADS ? ntiles : ntiles >= cap ? cap : (ntiles >= 8 ? (ntiles & ~7) : ntiles);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, s, a);
  VIDIL_CHECK_LAUNCH("gemm128x256");
  return VIDIL_OK;
}

template <typename T>
int dispatch_w4(const vidil_gemm_args& a, hipStream_t s) {
  if (a.ln_fold) {
    if (a.epi == VIDIL_EPI_HEADS) return launch_w4<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, true>(a, s);
    if (a.act == VIDIL_ACT_NONE) return launch_w4<T, VIDIL_EPI_F16, VIDIL_ACT_NONE, true>(a, s);
    if (a.act == VIDIL_ACT_GELU_ERF) return launch_w4<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF, true>(a, s);
    return launch_w4<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU, true>(a, s);
  }
  switch (a.epi) {
    case VIDIL_EPI_F16:
      if (a.act == VIDIL_ACT_NONE) return launch_w4<T, VIDIL_EPI_F16, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch_w4<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF>(a, s);
      return launch_w4<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_F32:
      if (a.ln_stats_out) return launch_w4<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true>(a, s);
      if (a.act == VIDIL_ACT_NONE) return launch_w4<T, VIDIL_EPI_F32, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch_w4<T, VIDIL_EPI_F32, VIDIL_ACT_GELU_ERF>(a, s);
      return launch_w4<T, VIDIL_EPI_F32, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_HEADS: return launch_w4<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE>(a, s);
    default: return launch_w4<T, VIDIL_EPI_PATCH, VIDIL_ACT_NONE>(a, s);
  }
}

template <typename TO>
int dispatch_w4_fp8(const vidil_gemm_args& a, hipStream_t s) {
  switch (a.epi) {
    case VIDIL_EPI_F32: return launch_w4<fp8, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, TO>(a, s);
    case VIDIL_EPI_PATCH: return launch_w4<fp8, VIDIL_EPI_PATCH, VIDIL_ACT_NONE, false, TO>(a, s);
    case VIDIL_EPI_HEADS: return launch_w4<fp8, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, false, TO>(a, s);
    case VIDIL_EPI_F8:
      if (a.act == VIDIL_ACT_GELU_ERF) return launch_w4<fp8, VIDIL_EPI_F8, VIDIL_ACT_GELU_ERF, false, TO>(a, s);
      if (a.act == VIDIL_ACT_QUICK_GELU) return launch_w4<fp8, VIDIL_EPI_F8, VIDIL_ACT_QUICK_GELU, false, TO>(a, s);
      return launch_w4<fp8, VIDIL_EPI_F8, VIDIL_ACT_NONE, false, TO>(a, s);
    default:
      vidil_set_error("gemm/fp8: epilogue %d is not built for fp8 operands", a.epi);
      return VIDIL_EUNSUP;
  }
}

}  // namespace

// The problem can run on the two-workgroups-per-CU kernel: fragment-tiled W supplied, enough tiles, and the vector
// epilogue rules of the 256x256 kernel (whose epilogue it shares).
bool vidil_gemm128x256_eligible(const vidil_gemm_args& a, bool any_size) {
  if (a.W_tiled == nullptr || ((uintptr_t)a.W_tiled & 15) != 0) return false;
  if (a.rln_gamma != nullptr) return false;           // the residual LayerNorm lives in the 256x256 kernel only
  if ((long)a.M * (a.lda > 0 ? a.lda : a.K) >= (1L << 31)) return false;   // this kernel's A offsets are 32-bit from p.A
  const long tiles = (long)((a.M + 127) / 128) * ((a.N + 255) / 256);
  if (tiles < 320 && !any_size) return false;
  return vidil_gemm256_eligible(a, true);
}

int vidil_gemm128x256_launch(const vidil_gemm_args& a, hipStream_t s) {
  if (a.dtype == VIDIL_DT_FP8) return a.dtype16 == VIDIL_DT_BF16 ? dispatch_w4_fp8<bf16>(a, s) : dispatch_w4_fp8<f16>(a, s);
  if (a.dtype == VIDIL_DT_BF16) return dispatch_w4<bf16>(a, s);
  return dispatch_w4<f16>(a, s);
}
