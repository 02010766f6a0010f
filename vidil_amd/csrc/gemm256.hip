// gemm256.hip — the large-M f16 GEMM of the hot path: C[M,N] = A[M,K] · W[N,K]^T
// with the same fused epilogues as gemm.hip, tiled 256x256x64 over 8 waves.
//
// Why a second kernel: a 128x128 tile needs 64 FLOP per byte staged from L2, i.e.
// ~39 TB/s of L2->LDS traffic at the 2.5 PFLOP/s MFMA peak — more than the 8 L2s
// deliver — so it tops out near 0.5 PFLOP/s.  256x256 halves the bytes per FLOP and
// the schedule below keeps the matrix pipe fed while LDS is being read.
//
// Structure (one workgroup = 512 threads = 8 waves = 2 per SIMD, 128 KiB LDS):
//   * LDS ring: 2 K-tiles x 4 half-tiles (A rows 0-127 / 128-255, W rows 0-127 /
//     128-255), 16 KiB each, filled by global_load_lds_dwordx4 (no VGPR round trip).
//     Rows are 128 B; the 16-B slot index is XORed with (row>>1)&7 on the SOURCE
//     address and on the ds_read_b128 side (conflict-free fragment reads).
//   * wave (g, wc): g = wave>>2 picks the A half (its 128 output rows), wc the 64
//     output columns.  The K-tile is processed in two HALF-PERIODS of 16 MFMAs
//     (v_mfma_f32_32x32x16_f16): output rows 0-63, then rows 64-127 of the wave.
//   * the two groups run STAGGERED by one half-period.  A group entering a K-tile
//     reads its 8 W fragments (kept in registers for both half-periods) and its
//     first A fragments from LDS before its first MFMA; at that moment the other
//     group (the other wave on every SIMD) is mid-tile with its W fragments
//     resident and its A fragments software-pipelined one k-step ahead, so the
//     matrix pipe does not wait for LDS.
//   * half-tiles are prefetched two half-periods ahead with COUNTED waits
//     (s_waitcnt vmcnt(2)/(6), never 0 in the loop) and raw s_barrier, so LDS-DMA
//     stays in flight across barriers.  Issue order per K-tile s: W0(s), W1(s),
//     A0(s), A1(s).  Even half-period 2v issues W0,W1,A0 of tile v+1; odd 2v+1
//     issues A1 of tile v+1.  (Derivation of the hazards: DESIGN.md §gemm256.)
//   * MFMA operands are swapped (D = W_frag · A_frag^T) so a lane ends up with 4
//     CONSECUTIVE output columns of one row; the epilogue transposes through the
//     (now idle) LDS so every global store is 16 B per lane and 128-256 B per row.
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int SLOT = 16384;      // one half-tile: 128 rows x 64 halfs
constexpr int BUF = 4 * SLOT;     // one K-tile: A0, A1, W0, W1
constexpr int LDS_BYTES = 2 * BUF;
constexpr int STATS_BYTES = 2 * 4 * 128 * 8;   // LN-fold consumers: [group][wave column][row] (sum, sum of squares)

__device__ __forceinline__ void glds16(const void* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// The per-head scatter epilogue (EPI_HEADS) is launched one tile per workgroup: its V^T tiles end in 128 narrow
// stores per lane whose drain the next tile would have to wait for, and the extra live state spills registers.
template <int EPI>
constexpr bool kPersistent = EPI != VIDIL_EPI_HEADS && EPI != VIDIL_EPI_ARENA;

#ifdef VIDIL_GEMM_PROBE
// developer build only (make EXTRA=-DVIDIL_GEMM_PROBE, tools/probe_gemm_clock.py): shader cycles and 100-MHz
// reference ticks workgroup 0 spent in its last launch -> the clock the kernel actually ran at
__device__ unsigned long long g_probe[2];
struct Probe {
  unsigned long long t0, r0;
  __device__ Probe() : t0(__builtin_readcyclecounter()), r0(__builtin_amdgcn_s_memrealtime()) {}
  __device__ ~Probe() {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g_probe[0] = __builtin_readcyclecounter() - t0;
      g_probe[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
  }
};
#endif

// FOLD: LayerNorm folded into this GEMM (vidil_gemm_args.ln_fold, include/vidil_hip.h).  A holds the RAW residual
// stream in the operand type; the producing GEMM left per-row partial sums / sums of squares (one pair per 64 columns,
// computed in its memory-bound f32 epilogue — statistics taken from the A fragments inside THIS main loop cost 6-8 % of
// the kernel: v_dot2c beside MFMAs).  The four waves that own the same rows each add up a quarter of the partials,
// meet in LDS after the main loop, and the epilogue applies  y = rstd * acc - (rstd * mean) * colsum[n] + b'[n].
// T: operand type of A and W (f16 / bf16 / fp8); TO: the 16-bit type of 16-bit outputs (== T unless T is fp8).
// RLN: the f32 residual is the RAW sum of a post-LN block and is normalised on the way in (vidil_gemm_args.rln_gamma):
// the same row statistics machinery as FOLD, applied to the residual rows instead of the accumulators.
template <typename T, typename TO, int EPI, int ACT, bool FOLD, bool STATS, bool RLN>
__global__ __launch_bounds__(512) void gemm256_kernel(const vidil_gemm_args p) {
  constexpr bool ROWSTAT = FOLD || RLN;          // per-row mean / rstd from a producer's partials
  static_assert(!(FOLD && RLN), "a GEMM normalises either its A rows or its residual rows");
  static_assert(!RLN || EPI == VIDIL_EPI_F32, "the residual exists in the f32 epilogue only");
  using f16 = TO;                       // (the epilogue is written in terms of "the 16-bit output type")
  using f16x4 = typename Elt<TO>::x4;
  using f16x8 = typename Elt<TO>::x8;
  using Frag = typename Mma<T>::Frag;
  constexpr int KS = Mma<T>::KS;
  constexpr int ESZ = sizeof(T);        // bytes per operand element
  constexpr int KT = 128 / ESZ;         // operand elements per K-tile (one 128-byte LDS row)
  static_assert(!FOLD || ESZ == 2, "the LayerNorm fold reads 16-bit A fragments");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VIDIL_GEMM_PROBE
  Probe probe;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // A half owned by this wave == stagger group
  const int wc = wave & 3;
  const int hi = lane >> 5, l31 = lane & 31;

  const int M = p.M, N = p.N, K = p.K;
  const int lda = p.lda > 0 ? p.lda : K;
  const int tiles_n = (N + 255) >> 8;
  const int tiles_m = (M + 255) >> 8;
  // PERSISTENT workgroups: the grid is (at most) one workgroup per CU; workgroup b lives on XCD b % 8 and walks
  // that XCD's contiguous range of logical tiles (consecutive tiles share an A row panel -> the XCD's L2) with a
  // stride of gridDim/8, so at any moment the 32 CUs of an XCD work on 32 consecutive tiles, as before.  What
  // persistence buys: the first K-tile of the NEXT tile is fetched while this tile's epilogue runs.
  int logical, remaining;
  const int tile_step = gridDim.x >= 8 ? (gridDim.x >> 3) : 1;   // (grids below 8 workgroups: one tile each, never 0)
  {
    const int nblk = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7, slot = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    remaining = (xcd < r ? q + 1 : q) - slot;     // tiles from `logical` to the end of the XCD's range
  }
  if (remaining <= 0) return;
  const int nk = K / KT;

  // ---- staging sources: half-tile hf of A / W, two 16-B chunks per thread ------------------------
  // (32-bit element offsets from the two uniform base pointers: 8 VGPRs instead of 16 for pointers — this
  //  kernel lives at the 256-register limit and a spilled pointer costs a vmcnt(0) drain per reload)
  int gA[2][2];
  int gW[2][2];
  int m0, n0;
  // tile order: row-panel-major (consecutive logical tiles share an A row panel and sweep all W column tiles; orders
  // that keep fewer column tiles of W live per XCD measured -13 ... +1 %)
  const T* baseA = (const T*)p.A;        // first row of the current tile (setup_tile)
  auto setup_tile = [&](int lt) {
    const int tile_m = lt / tiles_n;
    const int tile_n = lt - tile_m * tiles_n;
    m0 = tile_m << 8;
    n0 = tile_n << 8;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        const int r = q >> 3, sl = q & 7;
        const int c = sl ^ ((r >> 1) & 7);
        int ra = m0 + hf * 128 + r;
        ra = ra < M ? ra : M - 1;
        int rw = n0 + hf * 128 + r;
        rw = rw < N ? rw : N - 1;
        gA[hf][i] = (ra - m0) * lda + c * (16 / ESZ);     // relative to the tile's first row (baseA below): < 256 * lda
        gW[hf][i] = rw * K + c * (16 / ESZ);
      }
    baseA = (const T*)p.A + (size_t)m0 * lda;             // uniform: 64-bit once per tile, so M * lda may exceed 2^31
  };
  setup_tile(logical);
  const T* const baseW = (const T*)p.W;
  // slot ids inside a K-tile buffer: 0 = A0, 1 = A1, 2 = W0, 3 = W1.  Tiles past the end re-fetch the
  // last tile into the slot the schedule says is free, so the counted waits stay uniform.
  auto issue = [&](const T* base, const int(&g)[2], int tile, int slot) {
    const int tt = tile < nk ? tile : nk - 1;
    char* dst = smem + (tile & 1) * BUF + slot * SLOT + wave * 1024;
    const T* src = base + tt * KT;
    glds16(src + g[0], dst);
    glds16(src + g[1], dst + 8192);
  };

  f32x16 acc[4][2];
  float st_s[4], st_ss[4];   // FOLD: per row tile `it` — rstd / mean*rstd of the lane's row
  f32x2 st_raw[4][2];        // FOLD: this lane's share of the producer's (sum, sum of squares) row partials

  const int sw = (l31 >> 1) & 7;
  const int a_off = grp * SLOT + l31 * 128;
  const int w_off = (2 + (wc >> 1)) * SLOT + ((wc & 1) * 64 + l31) * 128;
  Frag wf[2][KS];   // W fragments of the current K-tile (both column tiles), resident for both half-periods

  const int nhp = 2 * nk;
  // DMA of GLOBAL half-period h, in three pieces so it can be interleaved with MFMAs (an LDS-DMA instruction
  // costs the issuing wave 100+ cycles of issue time; back to back at the top of a half-period they would
  // idle the matrix pipe): even h issues W0, W1, A0 of tile h/2+1, odd h issues A1 of tile (h-1)/2+1.
  auto dma_piece = [&](int h, int piece) {
    if (h >= nhp) return;
    const int v = (h >> 1) + 1;
    if (h & 1) {
      if (piece == 0) issue(baseA, gA[1], v, 1);
    } else {
      if (piece == 0) issue(baseW, gW[0], v, 2);
      if (piece == 1) issue(baseW, gW[1], v, 3);
      if (piece == 2) issue(baseA, gA[0], v, 0);
    }
  };
  // Start of GLOBAL half-period h (all 8 waves execute this together).  What h reads has landed once at
  // most 1 (even h: A1 of the same tile) or 3 (odd h: W0, W1, A0 of the next tile) newer half-tile DMAs
  // (2 instructions each) are still in flight; the barrier then makes every wave's DMA visible to every
  // wave and orders the previous half-period's LDS reads before this half-period's DMA reuses their slots.
  auto sync = [&](int h) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (h & 1) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  };

  // One half-period = 4 k-steps x (2 row tiles x 2 column tiles) MFMAs on output rows RH*64 .. RH*64+63 of
  // the wave; A fragments stream from LDS one k-step ahead of the MFMAs that consume them, and this
  // half-period's DMA pieces are issued behind the MFMAs of k-steps 0..2.
  auto half_period = [&](auto rh_tag, const char* buf, int h) {
    constexpr int RH = decltype(rh_tag)::value;
    const char* ab = buf + a_off + RH * 8192;
    // 16-bit operands: A fragments one k-step ahead.  fp8 fragments are 8-register tuples: one k-step's worth at a
    // time (the look-ahead made the allocator spill >2000 registers; the other wave of the SIMD covers the LDS latency)
    constexpr int NA = ESZ == 2 ? KS : 1;
    Frag a[NA][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[0][i] = Mma<T>::load(ab + i * 4096, 0, hi, sw);
    if constexpr (RH == 0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[j][ks] = Mma<T>::load(buf + w_off + j * 4096, ks, hi, sw);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (NA > 1) {
        if (ks < KS - 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i) a[ks + 1][i] = Mma<T>::load(ab + i * 4096, ks + 1, hi, sw);
        }
      } else if (ks > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[0][i] = Mma<T>::load(ab + i * 4096, ks, hi, sw);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[RH * 2 + i][j] = Mma<T>::mma(wf[j][ks], a[NA > 1 ? ks : 0][i], acc[RH * 2 + i][j]);
      // this half-period's (up to) three DMA pieces go out behind MFMAs, never right in front of the next barrier
      if constexpr (KS == 4) {
        if (ks < 3) dma_piece(h, ks);
      } else {
        if (ks == 0) { dma_piece(h, 0); dma_piece(h, 1); dma_piece(h, 2); }
      }
    }
  };

  // ---- prologue of a tile: W0(0), W1(0), A0(0), A1(0) into K-tile buffer 0 ------------------------------
  auto prologue = [&]() {
    issue(baseW, gW[0], 0, 2);
    issue(baseW, gW[1], 0, 3);
    issue(baseA, gA[0], 0, 0);
    issue(baseA, gA[1], 0, 1);
  };
  prologue();

  for (;;) {   // ======================================================================== one output tile
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[it][j][r] = 0.f;
  if constexpr (ROWSTAT) {
    // this wave's share of the producer's row partials (parts wc, wc+4, ...; the half-waves alternate): issued here,
    // consumed after the main loop
    // (loaded here, summed after the main loop: used right away they would expose a full memory latency per tile)
    const int nparts = (FOLD ? K : N) >> 6;   // <= 16 (check_args: the normalised rows are at most 1024 wide)
    const f32x2* stats_in = (const f32x2*)p.ln_stats;
    const int part0 = wc + 4 * hi;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int row = m0 + grp * 128 + it * 32 + l31;
      row = row < M ? row : M - 1;
      st_raw[it][0] = part0 < nparts ? stats_in[(size_t)row * nparts + part0] : f32x2{0.f, 0.f};
      st_raw[it][1] = part0 + 8 < nparts ? stats_in[(size_t)row * nparts + part0 + 8] : f32x2{0.f, 0.f};
    }
  }

  // Both groups run the same straight-line [first half, second half] body per K-tile (so the accumulators
  // never pass through a branch); group 1 simply starts one half-period later and group 0 idles in the last.
  int h = 0;
  if (grp == 1) {
    sync(h);
    dma_piece(h, 0); dma_piece(h, 1); dma_piece(h, 2);
    ++h;
  }
  for (int u = 0; u < nk; ++u) {
    const char* buf = smem + (u & 1) * BUF;
    sync(h);
    half_period(std::integral_constant<int, 0>{}, buf, h);
    ++h;
    sync(h);
    half_period(std::integral_constant<int, 1>{}, buf, h);
    ++h;
  }
  if (grp == 0) sync(h);   // h == nhp: nothing left to issue
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if constexpr (ROWSTAT) {
    // LayerNorm statistics of this tile's rows: every wave holds, per row, the sums over the k-steps it owns (and a
    // half-wave over its 8 of the step's 16 k) -> combine the half-waves, meet the other three waves of the group in
    // LDS (a region past the ring: nothing else touches it), then rstd and mean*rstd per row in fixed wave order.
    f32x2* stats = (f32x2*)(smem + LDS_BYTES);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float s0 = st_raw[it][0][0] + st_raw[it][1][0], ss0 = st_raw[it][0][1] + st_raw[it][1][1];
      const float s = s0 + __shfl_xor(s0, 32, 64);
      const float ss = ss0 + __shfl_xor(ss0, 32, 64);
      if (hi == 0) stats[(grp * 4 + wc) * 128 + it * 32 + l31] = f32x2{s, ss};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float inv_k = 1.0f / (float)(FOLD ? K : N);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const f32x2 v = stats[(grp * 4 + w) * 128 + it * 32 + l31];
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
  }

  // ================================================================================ epilogue
  // acc[it][j][rq*4+e]: row = m_w + it*32 + l31 ; col = n_w + j*32 + rq*8 + hi*4 + e
  const int m_w = m0 + grp * 128;
  const int n_w = n0 + wc * 64;
  // The next tile's first K-tile goes out NOW, into K-tile buffer 0 (every wave is past the barrier above, so
  // the ring is idle); the epilogue below transposes through buffer 1 only.  Its latency — an HBM miss on a new
  // A panel — is then covered by the epilogue instead of idling the CU at the top of the next tile.
  // (fp8 operands: one tile per workgroup too — the next tile's state across the epilogue costs a few spilled registers)
  const bool more = kPersistent<EPI> && ESZ == 2 && !RLN && remaining > tile_step;   // uniform (RLN: one tile per
  //                                                                                  workgroup — 18 spilled registers otherwise)
  if (more) {
    logical += tile_step;
    remaining -= tile_step;
    setup_tile(logical);
    prologue();
  }
  do {
    char* const ep = smem + BUF + wave * 8192;   // private 8-KiB transposition buffer of this wave, inside K-tile buffer 1
#include "gemm_epilogue.inc"
  } while (0);
  if (!more) break;
  // Before the next tile's counted waits: drain this tile's epilogue traffic (and, with it, the prologue issued
  // above — it has had the whole epilogue to land).  Loads and stores share vmcnt, and a counted wait is only
  // meaningful among the in-order LDS-DMA loads of the main loop.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }  // tile loop
}

template <typename T, int EPI, int ACT, bool FOLD = false, typename TO = T, bool STATS = false, bool RLN = false>
int launch256(const vidil_gemm_args& a, hipStream_t s) {
  static std::atomic<unsigned long long> attr_set{0};   // (one bit per device that has the opt-in: vidil_lds_opt_in)
  auto kern = gemm256_kernel<T, TO, EPI, ACT, FOLD, STATS, RLN>;
  constexpr int lds = LDS_BYTES + ((FOLD || RLN) ? STATS_BYTES : 0);
  if (const int rc_ = vidil_lds_opt_in(attr_set, (const void*)kern, lds, "gemm256")) return rc_;
  const int num_cu = vidil_cu_count() & ~7;     // (per device: core.hip)
  // persistent grid: one workgroup per CU (a multiple of 8 so every XCD gets the same number), never more than tiles
  int cus = num_cu;
  if (const char* e = vidil_dev_env("VIDIL_GEMM_CUS")) {      // developer: a stream confined to fewer CUs by a CU mask (tools/exp_cu_mask.py)
    const int v = atoi(e) & ~7;
    if (v >= 8 && v < cus) cus = v;
  }
  const int ntiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  const int tiles = !(kPersistent<EPI> && sizeof(T) == 2 && !RLN) ? ntiles : ntiles >= cus ? cus : (ntiles >= 8 ? ((ntiles + 7) & ~7) : ntiles);   // (rounded UP: gemm4w.hip launch4w)
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, s, a);
  VIDIL_CHECK_LAUNCH("gemm256");
  return VIDIL_OK;
}

}  // namespace

#ifdef VIDIL_GEMM_PROBE
extern "C" int vidil_debug_gemm_probe(unsigned long long* out2) {
  return hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_probe), 16) == hipSuccess ? 0 : -1;
}
#endif

// Returns true when the 256x256 kernel can run this problem with vector epilogues (checked by the caller).
// any_size: skip the "enough tiles to fill the chip" test (LN-folded GEMMs always run here, whatever M is).
bool vidil_gemm256_eligible(const vidil_gemm_args& a, bool any_size) {
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  if (tiles < 5L * vidil_cu_count() / 8 && !any_size) return false;      // (160 on 256 CUs) too few workgroups to fill the chip: small-tile kernel
  if (a.K < 128) return false;
  if (a.dtype == VIDIL_DT_FP8 && (a.K % 128 != 0 || (a.lda != 0 && a.lda % 16 != 0))) return false;
  const long lda = a.lda > 0 ? a.lda : a.K;
  if (256L * lda >= (1L << 31) || (long)a.N * a.K >= (1L << 31)) return false;   // 32-bit staging offsets (A: per row panel)
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  if (a.bias && !al16(a.bias)) return false;
  if (a.ln_fold && !(a.ln_colsum && al16(a.ln_colsum) && a.N % 4 == 0 && a.ln_stats && ((uintptr_t)a.ln_stats & 7) == 0)) return false;
  if (a.ln_stats_out && !(a.N % 64 == 0 && ((uintptr_t)a.ln_stats_out & 7) == 0)) return false;
  if (a.w_scale && !al16(a.w_scale)) return false;
  switch (a.epi) {
    case VIDIL_EPI_F8:
      return a.dtype == VIDIL_DT_FP8 && a.N % 16 == 0 && a.ldo % 16 == 0 && al16(a.out);
    case VIDIL_EPI_F16:
      return a.N % 8 == 0 && a.ldo % 8 == 0 && al16(a.out);
    case VIDIL_EPI_F32:
      if (a.out16 && !(a.ldo16 % 4 == 0 && ((uintptr_t)a.out16 & 7) == 0)) return false;
      return a.N % 4 == 0 && a.ldo % 4 == 0 && al16(a.out) && (!a.resid || al16(a.resid));
    case VIDIL_EPI_PATCH:
      return a.N % 4 == 0 && a.ldo % 4 == 0 && al16(a.out) && al16(a.pos);
    case VIDIL_EPI_HEADS:
      return (!a.q || al16(a.q)) && (!a.k || al16(a.k)) && ((a.NP != 0 && !a.kv_tiled) || !a.vt || al16(a.vt));
    case VIDIL_EPI_ARENA:      // (rows of H*64 16-bit values: 128-byte multiples)
      return (!a.q || al16(a.q)) && (!a.k || al16(a.k)) && (!a.vt || al16(a.vt));
    default:
      return false;
  }
}

template <typename T>
static int launch256_dispatch(const vidil_gemm_args& a, hipStream_t s) {
  if (a.ln_fold) {
    if (a.epi == VIDIL_EPI_HEADS) return launch256<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, true>(a, s);
    if (a.epi == VIDIL_EPI_ARENA) return launch256<T, VIDIL_EPI_ARENA, VIDIL_ACT_NONE, true>(a, s);   // (round 6: the decode steps' Q|K|V)
    if (a.act == VIDIL_ACT_NONE) return launch256<T, VIDIL_EPI_F16, VIDIL_ACT_NONE, true>(a, s);
    if (a.act == VIDIL_ACT_GELU_ERF) return launch256<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF, true>(a, s);
    return launch256<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU, true>(a, s);
  }
  switch (a.epi) {
    case VIDIL_EPI_F16:
      if (a.act == VIDIL_ACT_NONE) return launch256<T, VIDIL_EPI_F16, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch256<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF>(a, s);
      return launch256<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_F32:
      if (a.rln_gamma) return launch256<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true, true>(a, s);   // (check_args: with ln_stats_out)
      if (a.ln_stats_out) return launch256<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true>(a, s);   // (check_args: no activation)
      if (a.act == VIDIL_ACT_NONE) return launch256<T, VIDIL_EPI_F32, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch256<T, VIDIL_EPI_F32, VIDIL_ACT_GELU_ERF>(a, s);
      return launch256<T, VIDIL_EPI_F32, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_HEADS:
      return launch256<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE>(a, s);
    case VIDIL_EPI_ARENA:
      return launch256<T, VIDIL_EPI_ARENA, VIDIL_ACT_NONE>(a, s);
    default:
      return launch256<T, VIDIL_EPI_PATCH, VIDIL_ACT_NONE>(a, s);
  }
}

// fp8 operands: the residual / patch epilogues (f32 out), the per-head scatter into the 16-bit companion type, and the
// fp8 hand-over (fc1 -> fc2)
template <typename TO>
static int launch256_fp8(const vidil_gemm_args& a, hipStream_t s) {
  switch (a.epi) {
    case VIDIL_EPI_F32: return launch256<fp8, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, TO>(a, s);
    case VIDIL_EPI_PATCH: return launch256<fp8, VIDIL_EPI_PATCH, VIDIL_ACT_NONE, false, TO>(a, s);
    case VIDIL_EPI_HEADS: return launch256<fp8, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, false, TO>(a, s);
    case VIDIL_EPI_F8:
      if (a.act == VIDIL_ACT_GELU_ERF) return launch256<fp8, VIDIL_EPI_F8, VIDIL_ACT_GELU_ERF, false, TO>(a, s);
      if (a.act == VIDIL_ACT_QUICK_GELU) return launch256<fp8, VIDIL_EPI_F8, VIDIL_ACT_QUICK_GELU, false, TO>(a, s);
      return launch256<fp8, VIDIL_EPI_F8, VIDIL_ACT_NONE, false, TO>(a, s);
    default:
      vidil_set_error("gemm/fp8: epilogue %d is not built for fp8 operands", a.epi);
      return VIDIL_EUNSUP;
  }
}

int vidil_gemm4w_launch(const vidil_gemm_args& a, hipStream_t s, int tm);   // gemm4w.hip (tm: 4 = 256-row tiles, 2 = 128-row); -1000: variant not built there

// Which of the two 256x256 kernels runs a problem (both produce the same bits — tests/test_gemm4w_gpu.py).  The 4-wave
// kernel (gemm4w.hip) has the faster main loop (-13 % per K-tile) and a continuous K-tile stream, the 8-wave kernel hides
// memory latency in its epilogue with two waves per SIMD and starts up faster: measured on the same box inside the
// bench (profiles/r3_gemm4w_ab.md), gemm4w wins on the LN-folded consumers (-7 % on fc1 + GELU), the f32 + residual +
// row-partials producers (-4..5 %, with or without the residual LayerNorm) and the per-head scatter (-2..3 %) once there
// are a couple of tiles per CU, and loses on the plain f32 epilogue (+7 %) and on small grids (and is not built for fp8
// operands).  $VIDIL_GEMM4W = 0 / 1 forces one kernel (developer).
#ifndef VIDIL_GEMM4W_F32_DEFAULT
#define VIDIL_GEMM4W_F32_DEFAULT -1   // (-1: by K, see prefer_4w)
#endif
static bool prefer_4w(const vidil_gemm_args& a) {
  if (a.epi == VIDIL_EPI_HEADS && a.T < 8) return false;   // (the 4-wave scatter steps (image, token) by 8 rows: gemm_epilogue.inc)
  if (const char* e = vidil_dev_env("VIDIL_GEMM4W")) return atoi(e) != 0;
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  // (1.5 workgroup rounds: 384 tiles on 256 CUs)
  static const long min_tiles = [] { const char* e = getenv("VIDIL_GEMM4W_MIN_TILES"); return e ? atol(e) : 3L * vidil_cu_count() / 2; }();
  if (a.dtype == VIDIL_DT_FP8) {   // (round 4: the 4-wave main loop takes e4m3 operands too; same grid rule, plain epilogues;
    //                                $VIDIL_GEMM4W_FP8=0 keeps the fp8 GEMMs on the 8-wave kernel: the A/B of DESIGN.md §5)
    static const bool fp8_4w = [] { const char* e = getenv("VIDIL_GEMM4W_FP8"); return !(e && e[0] == '0'); }();
    return fp8_4w && tiles >= min_tiles &&
           (a.epi == VIDIL_EPI_F8 || a.epi == VIDIL_EPI_HEADS || (a.epi == VIDIL_EPI_F32 && a.act == VIDIL_ACT_NONE));
  }   // (developer sweep, whole bench, same box: 512 -> 5,112, 384 -> 5,120, 320 -> 5,094, 128 -> 5,050 frames/s — small grids start faster on gemm256)
  if (tiles < min_tiles) return false;
  if (a.ln_fold) return true;
  switch (a.epi) {
    case VIDIL_EPI_F16:
    case VIDIL_EPI_HEADS:
    case VIDIL_EPI_ARENA:
      return true;
    case VIDIL_EPI_F32: {
      // the LN-fold producers, with or without a residual LayerNorm: always; the plain f32 epilogue (the parity mode's K-tripled
      // GEMMs, the towers' last fc2, the LM head) by K: with 4 output bytes per 2K flop the f32 stores weigh on short reductions —
      // round 4, same box, whole bench: the last-block fc2 (K = 3072) 1,040 -> 1,160 TFLOP/s on the 4-wave kernel, the LM head
      // (10,752 x 30,524, K = 768) 838 -> 708.  $VIDIL_GEMM4W_F32 = 0 / 1 forces one kernel (A/B switch).
      static const int f32_4w = [] { const char* e = getenv("VIDIL_GEMM4W_F32"); return e ? atoi(e) : VIDIL_GEMM4W_F32_DEFAULT; }();
      // (round 5: with or without an activation — the parity mode's K-tripled fc1 + GELU, K = 2304, ran on the 8-wave kernel)
      return a.ln_stats_out != nullptr || f32_4w > 0 || (f32_4w < 0 && a.K >= 1536);
    }
    default:
      return false;
  }
}

const char* vidil_gemm256_variant(const vidil_gemm_args& a) { return prefer_4w(a) ? "gemm4w_kernel" : "gemm256_kernel"; }

// The 128 x 256-tile form of gemm4w for problems with too few 256 x 256 tiles to fill the chip but about one 128 x 256
// tile per CU or more (the decode steps' projections and FFN at ~10^4 beam rows): 16-bit operands, plain epilogues.
// Measured against the small-tile kernel at 10,752 rows (tools/experiments/gemm4w128_ab.py; all three kernels bit-identical):
// N = 768, K = 3072 (decode fc2) 61.7 -> 53.6 us, N = K = 768 with GELU / heads 23.4 -> 18.4, with the f32 residual 24.3 -> 24.0;
// where gemm256 is eligible (>= 160 tiles of 256 x 256) or K is a few tiles it loses, so it is not chosen there.
bool vidil_gemm4w128_wanted(const vidil_gemm_args& a) {
  const char* e = vidil_dev_env("VIDIL_GEMM4W128");                    // (developer: 0 never, 1 whenever it can run)
  const int mode = e ? atoi(e) : -1;
  if (mode == 0 || a.dtype == VIDIL_DT_FP8 || a.ln_fold || a.ln_stats_out || a.rln_gamma || a.out16) return false;
  if (a.epi == VIDIL_EPI_HEADS && a.T < 8) return false;
  if (!(a.epi == VIDIL_EPI_F16 || a.epi == VIDIL_EPI_HEADS || a.epi == VIDIL_EPI_ARENA || (a.epi == VIDIL_EPI_F32 && a.act == VIDIL_ACT_NONE)))
    return false;
  if (!vidil_gemm256_eligible(a, true)) return false;        // (alignment / stride / offset-range rules are the same)
  if (mode == 1) return true;
  const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 255) / 256);
  return t256 < 5L * vidil_cu_count() / 8 && t128 >= 25L * vidil_cu_count() / 32 && a.K >= 512;   // (160 / 200 on 256 CUs)
}
int vidil_gemm4w128_launch(const vidil_gemm_args& a, hipStream_t s) { return vidil_gemm4w_launch(a, s, 2); }

int vidil_gemm256_launch(const vidil_gemm_args& a, hipStream_t s) {
  if (prefer_4w(a)) {
    const int rc = vidil_gemm4w_launch(a, s, 4);
    if (rc != -1000) return rc;
  }
  if (a.dtype == VIDIL_DT_FP8) return a.dtype16 == VIDIL_DT_BF16 ? launch256_fp8<bf16>(a, s) : launch256_fp8<f16>(a, s);
  if (a.dtype == VIDIL_DT_BF16) return launch256_dispatch<bf16>(a, s);
  return launch256_dispatch<f16>(a, s);
}
