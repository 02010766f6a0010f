// gemm.hip — f16 MFMA GEMM  C[M,N] = A[M,K] · W[N,K]^T with fused epilogues.
//
// gfx950 design:
//   * 256-thread workgroup = 4 waves in a 2x2 grid; block tile BM x BN (128 or
//     64 each), K-step 64; every wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles
//     (v_mfma_f32_32x32x16_f16, f32 accumulate).
//   * A and W tiles go HBM -> LDS with global_load_lds_dwordx4 (16 B / lane,
//     no VGPR round trip), a 2- or 3-deep LDS ring with counted vmcnt waits
//     and one raw barrier per K-step.
//   * LDS rows are 128 B (64 halfs); the 16-B slot index is XOR-swizzled with
//     (row>>1)&7 on the SOURCE address (the LDS-DMA destination is lane
//     linear) and on the ds_read_b128 side, which makes the fragment reads
//     conflict free for the b128 lane groups.
//   * blockIdx is remapped so consecutive logical tiles (which share an A row
//     panel) run on one XCD and hit its private L2.
//   * epilogues: bias / GELU / residual / per-head Q-K-V^T scatter / patch
//     row remap + positional embedding, all on the f32 accumulators.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int BK = 64;
// LDS ring depth per tile shape
constexpr int ST_128x128 = 2;
constexpr int ST_128x64 = 2;
constexpr int ST_64x64 = 3;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// waves along N: the 128x128 tile runs on 8 waves (2 x 4, two per SIMD, 64x32 outputs each) so that one workgroup
// already overlaps LDS-DMA issue with MFMAs; the smaller tiles use 4 waves (2 x 2) and rely on co-resident workgroups
constexpr int waves_n(int bm, int bn) { return (bm == 128 && bn >= 128) ? 4 : 2; }

template <typename T, int BM, int BN, int ST, int EPI, int ACT>
__global__ __launch_bounds__(2 * waves_n(BM, BN) * 64) void gemm_kernel(const vidil_gemm_args p) {
  using f16 = T;                        // (the body is written in terms of "the 16-bit operand type")
  using f16x8 = typename Elt<T>::x8;
  constexpr int NWN = waves_n(BM, BN);
  constexpr int NT = 2 * NWN * 64;   // threads
  constexpr int WN = BN / NWN;       // output columns per wave
  constexpr int TM = BM / 64;  // 32x32 tiles per wave along M
  constexpr int TN = WN / 32;
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int PRE = ST - 1;  // K-tiles in flight ahead of the one being multiplied
  constexpr int LA = BM * 8 / NT;  // 16-B chunks per thread per stage (A)
  constexpr int LB = BN * 8 / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  const int M = p.M, N = p.N, K = p.K;
  const int lda = p.lda > 0 ? p.lda : K;
  const int tiles_n = (N + BN - 1) / BN;
  const int tiles_m = (M + BM - 1) / BM;
  // XCD-aware bijective remap of the block index (8 XCDs, block b -> XCD b%8).
  int logical;
  {
    const int nblk = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = logical / tiles_n;
  const int tile_n = logical - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;

  // per-thread source pointers for the staging loads (K offset added per step)
  const f16* ga[LA];
  const f16* gb[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int q = i * NT + tid;
    const int r = q >> 3, s = q & 7;
    const int c = s ^ ((r >> 1) & 7);
    int row = m0 + r;
    row = row < M ? row : M - 1;
    ga[i] = A + (size_t)row * lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int q = i * NT + tid;
    const int r = q >> 3, s = q & 7;
    const int c = s ^ ((r >> 1) & 7);
    int row = n0 + r;
    row = row < N ? row : N - 1;
    gb[i] = W + (size_t)row * K + c * 8;
  }

  auto stage = [&](int kt, int buf) {
    char* la = smem + buf * STAGE_BYTES;
    char* lb = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(ga[i] + kt * BK),
          (__attribute__((address_space(3))) void*)(la + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(gb[i] + kt * BK),
          (__attribute__((address_space(3))) void*)(lb + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sw = (lane >> 1) & 7;  // ((row>>1)&7) with row = 32*x + (lane&31)
  const int a_row_off = (wm * (BM / 2) + l31) * 128;
  const int b_row_off = (wn * WN + l31) * 128;

  // ST-deep ring: K-tile kt lives in buffer kt % ST and PRE tiles are in flight ahead of it, so the decode-step
  // GEMMs (one workgroup per CU, nothing else to hide the L2 latency behind) do not pay a full load latency
  // per K-step.  Waits are COUNTED (the LDS-DMA of the newer tiles stays in flight across the raw barrier);
  // the barrier both publishes every wave's DMA of tile kt and orders the reads of tile kt-1 before the DMA
  // that overwrites its buffer.
  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < PRE; ++s)
    if (s < nk) stage(s, s);
  int cur = 0, nxt = PRE;   // buffers of tile kt and of tile kt+PRE
  for (int kt = 0; kt < nk; ++kt) {
    const int newer = nk - 1 - kt;  // tiles issued after kt that may still be in flight (capped at PRE-1)
    if (newer >= PRE - 1) {
      wait_vm<(PRE - 1) * (LA + LB)>();
    } else if (PRE >= 3 && newer == 1) {
      wait_vm<LA + LB>();
    } else if (PRE >= 4 && newer == 2) {
      wait_vm<2 * (LA + LB)>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt + PRE < nk) stage(kt + PRE, nxt);
    const char* la = smem + cur * STAGE_BYTES;
    const char* lb = la + A_BYTES;
    cur = cur + 1 == ST ? 0 : cur + 1;
    nxt = nxt + 1 == ST ? 0 : nxt + 1;
    // all fragment reads of the K-tile are issued before its first MFMA (<= 64 VGPRs): with one wave per SIMD
    // nothing else hides the LDS latency, so it is paid once per K-tile instead of once per k-step
    f16x8 af[4][TM], bf[4][TN];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ((ks * 2 + hi) ^ sw) * 16;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[ks][i] = *(const f16x8*)(la + a_row_off + i * 32 * 128 + slot);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[ks][j] = *(const f16x8*)(lb + b_row_off + j * 32 * 128 + slot);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (the scheduler would re-sink them)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = Elt<T>::mfma32(af[ks][i], bf[ks][j], acc[i][j]);
  }

  // ---------------------------------------------------------------- epilogue
  // acc[i][j][r] : row = m0 + wm*BM/2 + i*32 + (r&3) + 8*(r>>2) + 4*hi
  //                col = n0 + wn*BN/2 + j*32 + (lane&31)
  // bias, then the activation in place on pairs of accumulators (packed-f32 instructions)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + j * 32 + l31;
    const float bias = (p.bias != nullptr && col < N) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 v = {acc[i][j][r] + bias, acc[i][j][r + 1] + bias};
        if constexpr (ACT == VIDIL_ACT_GELU_ERF) v = EPI == VIDIL_EPI_F32 ? gelu_erf2(v) : gelu_fast2<T>(v);   // (as gemm_epilogue.inc)
        if constexpr (ACT == VIDIL_ACT_QUICK_GELU) v = quick_gelu2(v);
        acc[i][j][r] = v[0];
        acc[i][j][r + 1] = v[1];
      }
  }
  // residual added up front: every load of the lane is in flight at once (inside the store loop each load
  // would have to wait for the previous store, because resid may alias out)
  if constexpr (EPI == VIDIL_EPI_F32) {
    if (p.resid != nullptr) {
      float rv[TM][TN][16];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            rv[i][j][r] = (col < N && row < M) ? p.resid[(size_t)row * p.ldo + col] : 0.f;
          }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += rv[i][j][r];
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + j * 32 + l31;
    const bool col_ok = col < N;
    // EPI_HEADS column decomposition
    int part = 0, hcol = 0;
    if constexpr (EPI == VIDIL_EPI_HEADS || EPI == VIDIL_EPI_ARENA) {
      const int hd = p.H * 64;
      part = p.part0 + col / hd;
      hcol = col % hd;  // h*64 + d
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int row_base = m0 + wm * (BM / 2) + i * 32 + 8 * rq + 4 * hi;
        int b = 0, t = 0;
        if constexpr (EPI == VIDIL_EPI_HEADS || EPI == VIDIL_EPI_ARENA) {
          b = row_base / p.T;
          t = row_base - b * p.T;
        } else if constexpr (EPI == VIDIL_EPI_PATCH) {
          b = row_base / p.tpi;
          t = row_base - b * p.tpi;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = row_base + rr;
          float v = acc[i][j][rq * 4 + rr];
          const bool ok = col_ok && row < M;
          if constexpr (EPI == VIDIL_EPI_F16) {
            if (ok) ((f16*)p.out)[(size_t)row * p.ldo + col] = Elt<T>::from_f32(v);
          } else if constexpr (EPI == VIDIL_EPI_F32) {
            if (ok) {
              ((float*)p.out)[(size_t)row * p.ldo + col] = v;
            }
          } else if constexpr (EPI == VIDIL_EPI_HEADS) {
            if (ok) {
              const int h = hcol >> 6, d = hcol & 63;
              const size_t bh = (size_t)b * p.H + h;
              if (part == 0) {
                ((f16*)p.q)[(bh * p.Tq_cap + t) * 64 + d] = Elt<T>::from_f32(v * p.q_scale);
              } else if (p.kv_tiled) {               // fragment tiles (common.h)
                const size_t base = bh * (size_t)p.Tk_cap * 64;
                if (part == 1) ((f16*)p.k)[base + ktile_off(p.t_off + t, d)] = Elt<T>::from_f32(v);
                else ((f16*)p.vt)[base + vtile_off(p.t_off + t, d)] = Elt<T>::from_f32(v);
              } else if (part == 1 || p.NP == 0) {   // NP == 0: V row-major, laid out like K
                ((f16*)(part == 1 ? p.k : p.vt))[(bh * p.Tk_cap + p.t_off + t) * 64 + d] = Elt<T>::from_f32(v);
              } else {
                ((f16*)p.vt)[(bh * 64 + d) * (size_t)p.NP + vt_pos(p.t_off + t)] = Elt<T>::from_f32(v);
              }
            }
            if (++t == p.T) { t = 0; ++b; }
          } else if constexpr (EPI == VIDIL_EPI_ARENA) {
            if (ok) {
              const size_t hd = (size_t)p.H * 64;
              if (part == 0) {
                ((f16*)p.q)[(size_t)row * hd + hcol] = Elt<T>::from_f32(v * p.q_scale);
              } else {
                f16* dst = (f16*)(part == 1 ? p.k : p.vt);
                dst[((size_t)(p.t_off + t) * p.arena_rows + (size_t)b * p.slot_stride) * hd + hcol] = Elt<T>::from_f32(v);
              }
            }
            if (++t == p.T) { t = 0; ++b; }
          } else {  // EPI_PATCH
            if (ok) {
              const size_t orow = (size_t)row + b + 1;
              ((float*)p.out)[orow * p.ldo + col] = v + p.pos[(size_t)(t + 1) * N + col];
            }
            if (++t == p.tpi) { t = 0; ++b; }
          }
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int ST, int EPI, int ACT>
int launch(const vidil_gemm_args& a, hipStream_t s) {
  constexpr int smem = ST * (BM + BN) * BK * 2;
  static std::atomic<unsigned long long> attr_set{0};   // (one bit per device that has the opt-in: vidil_lds_opt_in)
  auto kern = gemm_kernel<T, BM, BN, ST, EPI, ACT>;
  if (const int rc_ = vidil_lds_opt_in(attr_set, (const void*)kern, smem, "gemm")) return rc_;
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(2 * waves_n(BM, BN) * 64), smem, s, a);
  VIDIL_CHECK_LAUNCH("gemm");
  return VIDIL_OK;
}

// Which kernel serves a problem.  big != 0: the 256x256 kernel of gemm256.hip; else the small-tile kernel <bm, bn, st>.
struct TileChoice { int big, bm, bn, st; };

TileChoice choose_tile(const vidil_gemm_args& a) {
  static const bool allow256 = []() {
    const char* e = getenv("VIDIL_GEMM256");
    return !(e && e[0] == '0');
  }();
  const bool forced_big = a.ln_fold || a.out16 || a.ln_stats_out || a.dtype == VIDIL_DT_FP8;   // (check_args)
  if (forced_big) return {1, 256, 256, 2};
  if (allow256 && vidil_gemm4w128_wanted(a)) return {2, 128, 256, 2};
  if (allow256 && vidil_gemm256_eligible(a)) return {1, 256, 256, 2};
#ifdef VIDIL_GEMM_TUNE
  // developer builds only: VIDIL_GEMM_TILE=<BM>x<BN>x<ST> forces one configuration
  if (const char* e = vidil_dev_env("VIDIL_GEMM_TILE")) {
    int bm = 0, bn = 0, st = 0;
    if (sscanf(e, "%dx%dx%d", &bm, &bn, &st) == 3) return {0, bm, bn, st};
  }
#endif
  // Problems too small for gemm256 run one or two rounds of workgroups, so what matters is how many waves
  // share a SIMD (an LDS-DMA instruction costs its wave ~100 cycles of issue time that only another wave can
  // fill with MFMAs), not bytes per FLOP: grow the tile only once there are >= ~5 workgroups per CU.
  // (measured on the decode-step shapes M = 384..3072, tools/tune_gemm.py)
  // narrow outputs (N <= 1024: attention / cross-attention / FFN output projections of a decode step) with at least
  // ~200 128x128 tiles: the 8-wave 128x128 tile, 2-deep ring when two workgroups share a CU, 3-deep when alone
  // (M = 9216, K = 3072: 61 us against 74 us on 128x64; M = 4608: 33 against 43)
  const long n128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const long cus = vidil_cu_count();          // (the thresholds below are the tuned 200 / 256 / 1280 / 2560 on 256 CUs)
  if (n128 >= 25 * cus / 32 && a.N <= 1024) return {0, 128, 128, n128 > cus ? 2 : 3};
  const long n64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
  if (n64 <= 5 * cus) return {0, 64, 64, ST_64x64};
  const long n128x64 = (long)((a.M + 127) / 128) * ((a.N + 63) / 64);
  if (n128x64 <= 10 * cus) return {0, 128, 64, ST_128x64};
  return {0, 128, 128, ST_128x128};
}

template <typename T, int EPI, int ACT>
int pick_tile(const vidil_gemm_args& a, hipStream_t s) {
  const TileChoice c = choose_tile(a);
  if (c.big == 2) {
    const int rc = vidil_gemm4w128_launch(a, s);
    if (rc != -1000) return rc;
    vidil_set_error("gemm: the 128x256 kernel is not built for epilogue %d", a.epi);
    return VIDIL_EUNSUP;
  }
  if (c.big) return vidil_gemm256_launch(a, s);
#define VIDIL_TRY(BM_, BN_, ST_) \
  if (c.bm == BM_ && c.bn == BN_ && c.st == ST_) return launch<T, BM_, BN_, ST_, EPI, ACT>(a, s);
  VIDIL_TRY(128, 128, 2) VIDIL_TRY(128, 128, 3) VIDIL_TRY(128, 64, 2) VIDIL_TRY(64, 64, 3)
#ifdef VIDIL_GEMM_TUNE
  VIDIL_TRY(128, 128, 4) VIDIL_TRY(128, 64, 3) VIDIL_TRY(128, 64, 4) VIDIL_TRY(64, 64, 2) VIDIL_TRY(64, 64, 4)
  VIDIL_TRY(128, 256, 2) VIDIL_TRY(128, 256, 3)
#endif
#undef VIDIL_TRY
  vidil_set_error("gemm: no kernel for tile %dx%dx%d", c.bm, c.bn, c.st);
  return VIDIL_EUNSUP;
}

// argument checks shared by vidil_gemm and vidil_gemm_kernel_name
int check_args(const vidil_gemm_args& a) {
  VIDIL_REQUIRE(a.A && a.W, "gemm: null operand");
  VIDIL_REQUIRE(a.dtype == VIDIL_DT_F16 || a.dtype == VIDIL_DT_BF16 || a.dtype == VIDIL_DT_FP8, "gemm: unknown dtype %d", a.dtype);
  if (a.dtype == VIDIL_DT_FP8) {
    VIDIL_REQUIRE(a.K % 128 == 0, "gemm/fp8: K=%d must be a multiple of 128", a.K);
    VIDIL_REQUIRE(a.w_scale != nullptr, "gemm/fp8: w_scale (the per-output-column weight scale) is required");
    VIDIL_REQUIRE(a.dtype16 == VIDIL_DT_F16 || a.dtype16 == VIDIL_DT_BF16, "gemm/fp8: dtype16=%d must name the 16-bit output type", a.dtype16);
    VIDIL_REQUIRE(!a.ln_fold && !a.out16, "gemm/fp8: the LayerNorm fold is a 16-bit feature");
    VIDIL_REQUIRE(a.epi != VIDIL_EPI_F16 && a.epi != VIDIL_EPI_ARENA, "gemm/fp8: epilogue %d is not built for fp8 operands", a.epi);
  } else {
    VIDIL_REQUIRE(a.epi != VIDIL_EPI_F8, "gemm: EPI_F8 needs fp8 operands");
  }
  VIDIL_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
  VIDIL_REQUIRE(a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
  VIDIL_REQUIRE(a.lda == 0 || (a.lda >= a.K && a.lda % 8 == 0), "gemm: lda=%d must be 0 or >= K and a multiple of 8", a.lda);
  VIDIL_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: operands must be 16-B aligned");
  if (a.ln_fold) {
    VIDIL_REQUIRE(a.epi == VIDIL_EPI_F16 || a.epi == VIDIL_EPI_HEADS || a.epi == VIDIL_EPI_ARENA,
                  "gemm/ln_fold: only EPI_F16 / EPI_HEADS / EPI_ARENA consume a folded LayerNorm");
    VIDIL_REQUIRE(a.ln_colsum != nullptr && a.ln_stats != nullptr && a.ln_eps >= 0.f, "gemm/ln_fold: null ln_colsum / ln_stats");
    VIDIL_REQUIRE(a.lda == 0 || a.lda == a.K, "gemm/ln_fold: A rows must be dense (K = the LayerNorm width)");
    VIDIL_REQUIRE(a.K <= 1024, "gemm/ln_fold: LayerNorm widths up to 1024 (K=%d)", a.K);
  }
  if (a.out16) {
    VIDIL_REQUIRE(a.epi == VIDIL_EPI_F32, "gemm/out16: only the f32 residual epilogue writes the 16-bit copy");
    VIDIL_REQUIRE(a.ldo16 >= a.N, "gemm/out16: ldo16=%d < N=%d", a.ldo16, a.N);
  }
  if (a.out16_split3) {
    VIDIL_REQUIRE(a.out16_split3 == 1 || a.out16_split3 == 2, "gemm/out16_split3: 1 (three planes) or 2 (hi | lo only), got %d", a.out16_split3);
    VIDIL_REQUIRE(a.out16 && a.epi == VIDIL_EPI_F32 && !a.ln_stats_out && !a.rln_gamma && a.dtype != VIDIL_DT_FP8,
                  "gemm/out16_split3: the plain f32 epilogue with out16 (16-bit operands, no ln_stats_out / rln)");
    VIDIL_REQUIRE(a.ldo16 % 12 == 0 && a.ldo16 / 3 >= a.N && ((uintptr_t)a.out16 & 7) == 0,
                  "gemm/out16_split3: ldo16=%d must be three planes of >= N=%d columns, each a multiple of 4", a.ldo16, a.N);
  }
  VIDIL_REQUIRE(a.split_k >= 0 && a.split_k <= 2, "gemm/split_k: 0, 1 (three planes) or 2 (planes hi | lo only), got %d", a.split_k);
  if (a.split_k)
    VIDIL_REQUIRE(a.K % 3 == 0 && (a.K / 3) % 32 == 0 && a.dtype != VIDIL_DT_FP8 && !a.ln_fold,
                  "gemm/split_k: K=%d must be three planes of a multiple of 32 columns (16-bit operands, no ln_fold)", a.K);
  if (a.ln_stats_out)
    VIDIL_REQUIRE(a.epi == VIDIL_EPI_F32 && a.act == VIDIL_ACT_NONE && a.N % 64 == 0 && a.dtype != VIDIL_DT_FP8,
                  "gemm/ln_stats_out: 16-bit operands, f32 residual epilogue without activation, N %% 64 == 0");
  if (a.rln_gamma || a.rln_beta) {
    VIDIL_REQUIRE(a.rln_gamma && a.rln_beta && a.epi == VIDIL_EPI_F32 && a.resid && a.ln_stats && a.ln_stats_out && !a.ln_fold,
                  "gemm/rln: the residual LayerNorm needs rln_gamma, rln_beta, the f32 residual epilogue, resid, ln_stats (of resid) and ln_stats_out, and no ln_fold");
    VIDIL_REQUIRE(a.N % 64 == 0 && a.N <= 1024 && a.ldo == a.N, "gemm/rln: dense residual rows of width N %% 64 == 0, N <= 1024 (N=%d ldo=%d)", a.N, a.ldo);
    VIDIL_REQUIRE(((uintptr_t)a.rln_gamma & 15) == 0 && ((uintptr_t)a.rln_beta & 15) == 0 && ((uintptr_t)a.ln_stats & 7) == 0,
                  "gemm/rln: rln_gamma / rln_beta 16-byte aligned, ln_stats 8-byte aligned");
  }
  if (a.ln_fold || a.out16 || a.ln_stats_out || a.dtype == VIDIL_DT_FP8)
    VIDIL_REQUIRE(vidil_gemm256_eligible(a, true), "gemm: this LN-folded problem does not meet the 256x256 kernel's alignment / size rules (N %% 4, 16-B aligned vectors, K >= 128)");
  switch (a.epi) {
    case VIDIL_EPI_F16:
    case VIDIL_EPI_F32:
    case VIDIL_EPI_F8:
      VIDIL_REQUIRE((a.out || (a.epi == VIDIL_EPI_F32 && a.out16_split3 && !a.resid)) && a.ldo >= a.N, "gemm: bad out/ldo");
      VIDIL_REQUIRE(a.act >= VIDIL_ACT_NONE && a.act <= VIDIL_ACT_QUICK_GELU, "gemm: unknown act %d", a.act);
      return VIDIL_OK;
    case VIDIL_EPI_HEADS: {
      VIDIL_REQUIRE(a.act == VIDIL_ACT_NONE, "gemm/heads: no activation");
      VIDIL_REQUIRE(a.H > 0 && a.T > 0 && a.N % (a.H * 64) == 0, "gemm/heads: N=%d not a multiple of H*64 (H=%d)", a.N, a.H);
      const int nparts = a.N / (a.H * 64);
      VIDIL_REQUIRE(a.part0 >= 0 && a.part0 + nparts <= 3, "gemm/heads: part0=%d with %d parts", a.part0, nparts);
      for (int part = a.part0; part < a.part0 + nparts; ++part) {
        if (part == 0) VIDIL_REQUIRE(a.q && a.Tq_cap >= a.T, "gemm/heads: bad q / Tq_cap");
        if (part == 1) VIDIL_REQUIRE(a.k && a.Tk_cap >= a.t_off + a.T, "gemm/heads: bad k / Tk_cap");
        if (part >= 1 && a.kv_tiled) {
          VIDIL_REQUIRE((part == 1 || a.vt) && a.Tk_cap >= a.t_off + a.T && a.Tk_cap % 32 == 0,
                        "gemm/heads: tiled K/V need Tk_cap=%d >= t_off+T and a multiple of 32", a.Tk_cap);
          continue;
        }
        if (part == 2 && a.NP != 0)
          VIDIL_REQUIRE(a.vt && a.NP >= a.t_off + a.T && a.NP % 16 == 0, "gemm/heads: bad vt / NP (multiple of 16, >= t_off+T)");
        if (part == 2 && a.NP == 0)   // row-major V [b][h][Tk_cap][64]
          VIDIL_REQUIRE(a.vt && a.Tk_cap >= a.t_off + a.T, "gemm/heads: bad v / Tk_cap");
      }
      VIDIL_REQUIRE(a.M % a.T == 0, "gemm/heads: M=%d not a multiple of T=%d", a.M, a.T);
      return VIDIL_OK;
    }
    case VIDIL_EPI_ARENA: {
      VIDIL_REQUIRE(a.act == VIDIL_ACT_NONE, "gemm/arena: no activation");
      VIDIL_REQUIRE(a.H > 0 && a.T > 0 && a.N % (a.H * 64) == 0, "gemm/arena: N=%d not a multiple of H*64 (H=%d)", a.N, a.H);
      const int nparts = a.N / (a.H * 64);
      VIDIL_REQUIRE(a.part0 >= 0 && a.part0 + nparts <= 3, "gemm/arena: part0=%d with %d parts", a.part0, nparts);
      VIDIL_REQUIRE(a.M % a.T == 0, "gemm/arena: M=%d not a multiple of T=%d", a.M, a.T);
      VIDIL_REQUIRE(a.part0 > 0 || a.q, "gemm/arena: null q");
      if (a.part0 + nparts > 1) {
        VIDIL_REQUIRE(a.k && (a.part0 + nparts < 3 || a.vt), "gemm/arena: null k / v arena");
        VIDIL_REQUIRE(a.Tk_cap >= a.t_off + a.T, "gemm/arena: positions %d..%d exceed the arena capacity Tk_cap=%d", a.t_off,
                      a.t_off + a.T - 1, a.Tk_cap);
        VIDIL_REQUIRE(a.t_off >= 0 && a.slot_stride > 0 && (long)(a.M / a.T - 1) * a.slot_stride < a.arena_rows,
                      "gemm/arena: %d sequences at slot stride %d do not fit %d arena rows", a.M / a.T, a.slot_stride,
                      a.arena_rows);
      }
      return VIDIL_OK;
    }
    case VIDIL_EPI_PATCH:
      VIDIL_REQUIRE(a.act == VIDIL_ACT_NONE, "gemm/patch: no activation");
      VIDIL_REQUIRE(a.out && a.pos && a.tpi > 0 && a.M % a.tpi == 0 && a.ldo >= a.N, "gemm/patch: bad args");
      return VIDIL_OK;
    default:
      break;
  }
  vidil_set_error("gemm: unsupported epi=%d act=%d", a.epi, a.act);
  return VIDIL_EUNSUP;
}

template <typename T>
int dispatch(const vidil_gemm_args& a, hipStream_t s) {
  switch (a.epi) {
    case VIDIL_EPI_F16:
      if (a.act == VIDIL_ACT_NONE) return pick_tile<T, VIDIL_EPI_F16, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return pick_tile<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF>(a, s);
      return pick_tile<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_F32:
      if (a.act == VIDIL_ACT_NONE) return pick_tile<T, VIDIL_EPI_F32, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return pick_tile<T, VIDIL_EPI_F32, VIDIL_ACT_GELU_ERF>(a, s);
      return pick_tile<T, VIDIL_EPI_F32, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_HEADS: return pick_tile<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE>(a, s);
    case VIDIL_EPI_ARENA: return pick_tile<T, VIDIL_EPI_ARENA, VIDIL_ACT_NONE>(a, s);
    default: return pick_tile<T, VIDIL_EPI_PATCH, VIDIL_ACT_NONE>(a, s);
  }
}

}  // namespace

// split_k (the parity precision mode's operands, see include/vidil_hip.h): which launches run the in-loop compensated product
// (gemm4w's C3 form) — a property of the CALL (epilogue, alignment), never of its size, so that a row's result does not depend
// on the batch around it; everything else runs the same operands as a plain GEMM with K = 3 Kl (the K-tripled form of rounds 3-4).
static bool c3_switched_off() {
  static const bool off = [] { const char* e = getenv("VIDIL_GEMM_C3"); return e && e[0] == '0'; }();    // (A/B switch)
  return off;
}
extern "C" int vidil_gemm_split_k_in_loop(void) { return c3_switched_off() ? 0 : 1; }

bool vidil_gemm_c3_serves(const vidil_gemm_args& a) {
  if (!a.split_k || c3_switched_off()) return false;
  if (a.dtype == VIDIL_DT_FP8 || a.ln_fold || a.ln_stats_out || a.rln_gamma || a.K % 96 != 0) return false;
  if (!(a.epi == VIDIL_EPI_F32 || a.epi == VIDIL_EPI_PATCH || (a.epi == VIDIL_EPI_HEADS && a.T >= 8))) return false;
  return vidil_gemm256_eligible(a, true);
}

extern "C" int vidil_gemm_split_k_serves(const vidil_gemm_args* args) {
  VIDIL_REQUIRE(args != nullptr, "gemm_split_k_serves: null args");
  vidil_gemm_args a = *args;
  if (a.split_k == 2) a.split_k = 1;          // (the question is about the call, not about what the caller already assumed)
  const int rc = check_args(a);
  if (rc != VIDIL_OK) return rc;
  return vidil_gemm_c3_serves(a) ? 1 : 0;
}

extern "C" int vidil_gemm(const vidil_gemm_args* args, void* stream) {
  VIDIL_REQUIRE(args != nullptr, "gemm: null args");
  const int rc = check_args(*args);
  if (rc != VIDIL_OK) return rc;
  if (vidil_gemm_c3_serves(*args)) {
    const int rc3 = vidil_gemm4w_c3_launch(*args, (hipStream_t)stream);
    if (rc3 != -1000) return rc3;
  }
  if (args->split_k == 2) {
    // the caller wrote planes hi | lo of the A rows only: the plain K = 3 Kl product below would read plane 2
    vidil_set_error("gemm/split_k=2: this call does not qualify for the K-loop compensated product (epi=%d T=%d K=%d, alignment, or "
                    "$VIDIL_GEMM_C3=0) and its A rows hold planes hi | lo only — produce three planes and pass split_k=1", args->epi,
                    args->T, args->K);
    return VIDIL_EINVAL;
  }
  if (args->dtype == VIDIL_DT_FP8)
    return vidil_gemm256_launch(*args, (hipStream_t)stream);
  if (args->dtype == VIDIL_DT_BF16) return dispatch<bf16>(*args, (hipStream_t)stream);
  return dispatch<f16>(*args, (hipStream_t)stream);
}

extern "C" int vidil_gemm_kernel_name(const vidil_gemm_args* args, char* buf_host, int32_t n) {
  VIDIL_REQUIRE(args != nullptr && buf_host != nullptr && n > 0, "gemm_kernel_name: bad args");
  const int rc = check_args(*args);
  if (rc != VIDIL_OK) return rc;
  const TileChoice c = choose_tile(*args);
  const char* t16 = (args->dtype == VIDIL_DT_FP8 ? args->dtype16 : args->dtype) == VIDIL_DT_BF16 ? "__bf16" : "_Float16";
  if (vidil_gemm_c3_serves(*args)) {
    const long t256 = (long)((args->M + 255) / 256) * ((args->N + 255) / 256);
    const int tm = (t256 >= 5L * vidil_cu_count() / 8 || args->epi == VIDIL_EPI_PATCH) ? 4 : 2;
    snprintf(buf_host, n, "gemm4w_kernel<%s, %s, %d, %d, false, false, false, %d, true>", t16, t16, args->epi,
             args->epi == VIDIL_EPI_F32 ? args->act : 0, tm);
    return VIDIL_OK;
  }
  const char* t = args->dtype == VIDIL_DT_FP8 ? "fp8" : t16;                 // the spelling rocprofv3 demangles to
  const int act = (args->epi == VIDIL_EPI_F16 || args->epi == VIDIL_EPI_F32 || args->epi == VIDIL_EPI_F8) ? args->act : 0;
  const char* stats = (args->ln_stats_out && args->epi == VIDIL_EPI_F32) ? "true" : "false";
  if (c.big) {
    const char* kn = c.big == 2 ? "gemm4w_kernel" : vidil_gemm256_variant(*args);
    if (kn[4] == '4') snprintf(buf_host, n, "%s<%s, %s, %d, %d, %s, %s, %s, %d, false>", kn, t, t16, args->epi, act, args->ln_fold ? "true" : "false",
                               stats, args->rln_gamma ? "true" : "false", c.big == 2 ? 2 : 4);      // (gemm4w: + its row-tile count and the C3 flag)
    else snprintf(buf_host, n, "%s<%s, %s, %d, %d, %s, %s, %s>", kn, t, t16, args->epi, act, args->ln_fold ? "true" : "false", stats,
                  args->rln_gamma ? "true" : "false");
  }
  else snprintf(buf_host, n, "gemm_kernel<%s, %d, %d, %d, %d, %d>", t, c.bm, c.bn, c.st, args->epi, act);
  return VIDIL_OK;
}
