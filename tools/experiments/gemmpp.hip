// gemmpp.hip — the large-M GEMM of the hot path as a PING-PONG of two wave groups: C[M,N] = A[M,K] · W[N,K]^T with
// the fused epilogues of gemm256.hip, restructured so that the epilogue of one output tile runs UNDER the main loop of
// the next one.
//
// Why: at the K of this path (768 for three of the four tower GEMMs) a 256x256 tile spends 36 k cycles in its main loop
// and 8 k (16-bit store) / 22 k (erf-GELU) / 29 k (f32 residual read-modify-write) in its epilogue with the matrix pipe
// idle — SQ_VALU_MFMA_BUSY 0.40 on the dominant kernel (profiles/r2_bench_pmc.md).  A second accumulator set for the
// same wave does not fit the register file (256x256 f32 = half of it), a second workgroup per CU doubles the W traffic
// (tools/experiments/gemm128x256.hip, measured slower).  What does fit: ONE workgroup of 8 waves = two GROUPS of four
// (one wave per SIMD each) that take turns:
//
//      phase p     group p%2      : MAIN LOOP of output tile p (256 x 128; wave = 128 x 64, 128 accumulator registers),
//                                   the only MFMA stream on its SIMD — fragments double-buffered one k-step ahead
//                  group (p+1)%2  : EPILOGUE of tile p-1 from its own accumulators, cut into 8 PIECES (row tile it x
//                                   column tile j of the wave's 4 x 2 MFMA tiles) spread over the K-tiles of the phase,
//                                   AND the LDS-DMA ISSUE for the main group (an LDS-DMA instruction costs its wave
//                                   ~100 issue cycles: paid by the wave that has them to spare)
//
// so every SIMD always holds one MFMA-bound and one VALU / memory-bound wave — the pairing the hardware arbitrates best
// (MI355X_MICROARCH.md "two waves per SIMD") — and the matrix pipe only idles in the first and last phase of a workgroup.
// Both groups run the same program, offset by one phase; a wave's accumulators simply stay in its registers from its
// main phase into its epilogue phase.
//
// Ring and synchronisation (one s_barrier per K-tile, shared by all 8 waves):
//   * LDS: 2 stages x 3 slots (A rows 0-127, A rows 128-255, W rows 0-127; 16 KiB each, 128-B rows, 16-B slot index
//     XORed with (row>>1)&7 on the source address and on the ds_read_b128 side) + an 8-KiB transposition scratch per
//     wave of the group that is in its epilogue phase = 128 KiB.
//   * stream element s = (tile, K-tile) lives in stage s & 1.  Main wave, element s: k-steps 0 .. KS-2 (fragments of step
//     c+1 are read at the top of step c), lgkmcnt(0) — every read of element s has returned —, BARRIER B(s+1), last
//     k-step (whose look-ahead reads are the first fragments of element s+1).  Issuer wave, same period: issue the DMA
//     of element s+1 into the stage B(s) freed, run its epilogue piece, vmcnt(0) — its DMA has landed, its stores
//     are out —, BARRIER B(s+1).  So B(s+1) publishes element s+1 and frees the stage of element s; nothing is read before
//     the barrier after the wait that retires it.
// Results are bit-identical to gemm256.hip and gemm.hip (same k order per output element, same epilogue arithmetic).
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int PP_SLOT = 16384;                    // 128 rows x 128 B
constexpr int PP_STAGE = 3 * PP_SLOT;             // A0, A1, W
// Ring depth: 3 stages (144 KiB) where the epilogue's transposition scratch fits 4 KiB per wave (16-bit / fp8 rows: the
// whole 160 KiB of the CU), else 2 stages (96 KiB) + 8 KiB per wave.  With 3 stages the DMA of an element is issued TWO
// periods before the barrier that publishes it and the issuer never waits for what it has just issued.
template <int EPI> constexpr int kStages = (EPI == VIDIL_EPI_F16 || EPI == VIDIL_EPI_F8) ? 3 : 2;
template <int EPI> constexpr int kScratch = kStages<EPI> == 3 ? 4096 : 8192;          // per wave of the epilogue group
template <int EPI> constexpr int kLds = kStages<EPI> * PP_STAGE + 4 * kScratch<EPI>;  // 163840 / 131072

__device__ __forceinline__ void pp_glds16(const void* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <typename T, typename TO, int EPI, int ACT, bool FOLD, bool STATS, bool RLN>
__global__ __launch_bounds__(512) void gemmpp_kernel(const vidil_gemm_args p) {
  constexpr bool ROWSTAT = FOLD || RLN;
  static_assert(!(FOLD && RLN), "a GEMM normalises either its A rows or its residual rows");
  static_assert(!RLN || EPI == VIDIL_EPI_F32, "the residual exists in the f32 epilogue only");
  using f16 = TO;
  using f16x4 = typename Elt<TO>::x4;
  using f16x8 = typename Elt<TO>::x8;
  using Frag = typename Mma<T>::Frag;
  constexpr int KS = Mma<T>::KS;
  constexpr int ESZ = sizeof(T);
  constexpr int KT = 128 / ESZ;
  constexpr int NST = kStages<EPI>;
  constexpr int PP_RING = NST * PP_STAGE;
  constexpr int PP_SCRATCH = kScratch<EPI>;
  static_assert(!FOLD || ESZ == 2, "the LayerNorm fold reads 16-bit A fragments");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;          // 0: main loop in even phases; 1: in odd phases
  const int wi = wave & 3;            // wave inside the group (one per SIMD)
  const int wr = wi >> 1, wc = wi & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int M = p.M, N = p.N, K = p.K;
  const int lda = p.lda > 0 ? p.lda : K;
  const int tiles_n = (N + 127) >> 7;
  const int tiles_m = (M + 255) >> 8;
  // persistent workgroups, XCD-aware as in gemm256: workgroup b lives on XCD b % 8 and walks that XCD's contiguous
  // range of logical tiles (row-panel-major: consecutive tiles share an A row panel) with a stride of gridDim / 8
  int first, nt;
  const int tile_step = gridDim.x >= 8 ? (gridDim.x >> 3) : 1;
  {
    const int nblk = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7, slot = bid >> 3;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int left = (xcd < r ? q + 1 : q) - slot;
    nt = left > 0 ? (left + tile_step - 1) / tile_step : 0;
  }
  if (nt <= 0) return;
  const int nk = K / KT;
  auto tile_origin = [&](int i, int& m0, int& n0) {
    const int lt = first + i * tile_step;
    const int tm = lt / tiles_n;
    m0 = tm << 8;
    n0 = (lt - tm * tiles_n) << 7;
  };

  // ---- LDS-DMA of one stream element (issuer group: 256 threads, 4 x 1 KiB per slot per wave) ------------------------
  int dA[2][4], dW[4];
  const T* dbaseA = (const T*)p.A;
  const T* const baseW = (const T*)p.W;
  auto dma_setup = [&](int i) {
    int m0, n0;
    tile_origin(i, m0, n0);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int q = ii * 256 + (tid & 255);
      const int r = q >> 3, sl = q & 7;
      const int c = sl ^ ((r >> 1) & 7);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        int ra = m0 + hf * 128 + r;
        ra = ra < M ? ra : M - 1;
        dA[hf][ii] = (ra - m0) * lda + c * (16 / ESZ);       // relative to the tile's first row: < 256 * lda
      }
      int rw = n0 + r;
      rw = rw < N ? rw : N - 1;
      dW[ii] = rw * K + c * (16 / ESZ);
    }
    dbaseA = (const T*)p.A + (size_t)m0 * lda;
  };
  auto dma_issue = [&](int kt, int stage) {
#if defined(VIDIL_PP_ABLATE) && (VIDIL_PP_ABLATE == 1)      // developer ablation (tools/exp_gemmpp.sh): no global traffic
    return;
#endif
    char* dst = smem + stage * PP_STAGE + wi * 1024;
    const T* srcA = dbaseA + kt * KT;
    const T* srcW = baseW + kt * KT;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) pp_glds16(srcA + dA[0][ii], dst + ii * 4096);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) pp_glds16(srcA + dA[1][ii], dst + PP_SLOT + ii * 4096);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) pp_glds16(srcW + dW[ii], dst + 2 * PP_SLOT + ii * 4096);
  };

  f32x16 acc[4][2];
  float st_s[4], st_ss[4];    // ROWSTAT: rstd and mean * rstd of the lane's row, per row tile

  const int sw = (l31 >> 1) & 7;
  const int a_off = wr * PP_SLOT + l31 * 128;
  const int w_off = 2 * PP_SLOT + (wc * 64 + l31) * 128;

  // ================================================================================= main phase of tile q
  auto main_phase = [&](int q) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[it][j][r] = 0.f;
    int st = (q * nk) % NST;
    Frag a[2][4], w[2][2];
    auto load_set = [&](auto set_tag, const char* buf, int ks) {
      constexpr int S = decltype(set_tag)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j) w[S][j] = Mma<T>::load(buf + w_off + j * 4096, ks, hi, sw);
#pragma unroll
      for (int it = 0; it < 4; ++it) a[S][it] = Mma<T>::load(buf + a_off + it * 4096, ks, hi, sw);
    };
    load_set(std::integral_constant<int, 0>{}, smem + st * PP_STAGE, 0);
    __builtin_amdgcn_s_setprio(1);
    for (int j = 0; j < nk; ++j) {
      const int nst = st + 1 == NST ? 0 : st + 1;
      const char* buf = smem + st * PP_STAGE;
      const char* nbuf = smem + nst * PP_STAGE;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // (the scheduler must not sink the look-ahead reads below the MFMAs they are meant to run under: with one MFMA
        //  wave per SIMD nothing else covers an exposed LDS latency)
        __builtin_amdgcn_sched_barrier(0);
        // look-ahead: the fragments of the next k-step (after the barrier: of the next element) into the other set
        if (ks < KS - 1) {
          if (ks & 1) load_set(std::integral_constant<int, 0>{}, buf, ks + 1);
          else load_set(std::integral_constant<int, 1>{}, buf, ks + 1);
        } else if (j + 1 < nk) {
          if (ks & 1) load_set(std::integral_constant<int, 0>{}, nbuf, 0);
          else load_set(std::integral_constant<int, 1>{}, nbuf, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#if defined(VIDIL_PP_ABLATE) && (VIDIL_PP_ABLATE == 2)      // developer ablation: fragment reads and barriers, no MFMA
        asm volatile("" :: "v"(w[ks & 1][0]), "v"(w[ks & 1][1]), "v"(a[ks & 1][0]), "v"(a[ks & 1][1]), "v"(a[ks & 1][2]), "v"(a[ks & 1][3]));
#else
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[it][jj] = Mma<T>::mma(w[ks & 1][jj], a[ks & 1][it], acc[it][jj]);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // lgkmcnt(0) as the BUILTIN (simm16 0xC07F: vmcnt 63, expcnt 7, lgkmcnt 0): the compiler's own wait-count tracker
        // sees it — behind an inline-asm wait it re-waits, right after the next look-ahead reads were issued
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (ks == KS - 2) __builtin_amdgcn_s_barrier();
      }
      st = nst;
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ================================================================================= epilogue pieces
  int m_w = 0, n_w = 0;               // first row / column of this wave's 128 x 64 part of the tile being finished
  char* const ep = smem + PP_RING + wi * PP_SCRATCH;
  auto value = [&](int it, int j, int rq, int e) { return acc[it][j][rq * 4 + e]; };

  // row statistics of row tile IT from the producer's partials — the summation order of gemm256 (bit-identical):
  // for w = 0..3: (part[w] + part[w+8]) + (part[w+4] + part[w+12]), added up in that order
  auto row_stats = [&](auto it_tag) {
    constexpr int IT = decltype(it_tag)::value;
    const int nparts = (FOLD ? K : N) >> 6;
    const f32x2* stats_in = (const f32x2*)p.ln_stats;
    int row = m_w + IT * 32 + l31;
    row = row < M ? row : M - 1;
    f32x2 raw[4][2];
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
      const int p0 = w4 + 4 * hi;
      raw[w4][0] = p0 < nparts ? stats_in[(size_t)row * nparts + p0] : f32x2{0.f, 0.f};
      raw[w4][1] = p0 + 8 < nparts ? stats_in[(size_t)row * nparts + p0 + 8] : f32x2{0.f, 0.f};
    }
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
      const float s0 = raw[w4][0][0] + raw[w4][1][0], ss0 = raw[w4][0][1] + raw[w4][1][1];
      s += s0 + __shfl_xor(s0, 32, 64);
      ss += ss0 + __shfl_xor(ss0, 32, 64);
    }
    const float inv_k = 1.0f / (float)(FOLD ? K : N);
    const float mean = s * inv_k;
    float var = ss * inv_k - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = 1.0f / sqrtf(var + p.ln_eps);
    st_s[IT] = rstd;
    st_ss[IT] = mean * rstd;
  };

  // piece PI = (row tile IT = PI / 2, column tile J = PI % 2): LayerNorm fold / weight scale / bias and the activation
  // on the 16 accumulators of that MFMA tile; the odd piece then stores the wave's 32 x 64 block of row tile IT.
  auto piece = [&](auto pi_tag) {
    constexpr int PI = decltype(pi_tag)::value;
    constexpr int IT = PI >> 1, J = PI & 1;
    if (n_w >= N) return;
    if constexpr (ROWSTAT && J == 0) row_stats(std::integral_constant<int, IT>{});
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int col = n_w + J * 32 + rq * 8 + hi * 4;
      if (col + 4 <= N) {
        if constexpr (FOLD) {
          const f32x4 c4 = *(const f32x4*)(p.ln_colsum + col);
          f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
          if (p.bias != nullptr) b4 = *(const f32x4*)(p.bias + col);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[IT][J][rq * 4 + e] = __builtin_fmaf(acc[IT][J][rq * 4 + e], st_s[IT], __builtin_fmaf(-st_ss[IT], c4[e], b4[e]));
        } else if constexpr (ESZ == 1) {
          const f32x4 w4 = *(const f32x4*)(p.w_scale + col);
          f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
          if (p.bias != nullptr) b4 = *(const f32x4*)(p.bias + col);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[IT][J][rq * 4 + e] = __builtin_fmaf(acc[IT][J][rq * 4 + e], w4[e], b4[e]);
        } else {
          if (p.bias != nullptr) {
            const f32x4 b4 = *(const f32x4*)(p.bias + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[IT][J][rq * 4 + e] += b4[e];
          }
        }
      }
    }
    if constexpr (ACT != VIDIL_ACT_NONE) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 v = {acc[IT][J][r], acc[IT][J][r + 1]};
        v = ACT == VIDIL_ACT_GELU_ERF ? gelu_erf2(v) : quick_gelu2(v);
        acc[IT][J][r] = v[0];
        acc[IT][J][r + 1] = v[1];
      }
    }
    if constexpr (J == 0) return;

    // ------------------------------------------------------------------ store row tile IT (32 rows x 64 columns)
    int part = 0, head = 0;
    if constexpr (EPI == VIDIL_EPI_HEADS) {
      const int hd = p.H * 64;
      part = p.part0 + n_w / hd;
      head = (n_w % hd) >> 6;
      if (part == 2 && p.kv_tiled) {
        // V in fragment tiles (common.h vtile_off): [key][d] through the scratch (rows padded to 136 B), lane d then
        // collects runs of 4 tile-aligned keys of one image as 8-B stores (see gemm_epilogue.inc)
        constexpr int ROWB = 136;
        const size_t img_stride = (size_t)p.H * p.Tk_cap * 64;
        f16* const vbase = (f16*)p.vt + (size_t)head * p.Tk_cap * 64;
        const int mb = m_w + IT * 32;
        const int rows = M - mb < 32 ? M - mb : 32;   // wave-uniform
        if (rows <= 0) return;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const f16x4 v = {Elt<TO>::from_f32(value(IT, j, rq, 0)), Elt<TO>::from_f32(value(IT, j, rq, 1)),
                             Elt<TO>::from_f32(value(IT, j, rq, 2)), Elt<TO>::from_f32(value(IT, j, rq, 3))};
            *(f16x4*)(ep + l31 * ROWB + (j * 32 + rq * 8 + hi * 4) * 2) = v;
          }
        int b = mb / p.T, t = mb - b * p.T;
        for (int r = 0; r < rows;) {
          const int tt = p.t_off + t;
          f16* dst = vbase + (size_t)b * img_stride + vtile_off(tt, lane);
          const char* src = ep + r * ROWB + lane * 2;
          if ((tt & 3) == 0 && r + 4 <= rows && t + 4 <= p.T) {
            const f16x4 v = {*(const f16*)src, *(const f16*)(src + ROWB), *(const f16*)(src + 2 * ROWB),
                             *(const f16*)(src + 3 * ROWB)};
            *(f16x4*)dst = v;
            r += 4;
            t += 4;
          } else {
            *dst = *(const f16*)src;
            r += 1;
            t += 1;
          }
          if (t >= p.T) { t -= p.T; ++b; }
        }
        return;
      }
      if (part == 2 && p.NP != 0) {
        // V^T: element (row m, column d) goes to VT[b][h][d][t_off+t]; consecutive lanes = consecutive t
        const int m = m_w + IT * 32 + l31;
        if (m < M) {
          const int b = m / p.T, t = m - b * p.T;
          f16* dst = (f16*)p.vt + (((size_t)b * p.H + head) * 64) * (size_t)p.NP + vt_pos(p.t_off + t);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
#pragma unroll
              for (int e = 0; e < 4; ++e) dst[(size_t)(j * 32 + rq * 8 + hi * 4 + e) * p.NP] = Elt<TO>::from_f32(value(IT, j, rq, e));
        }
        return;
      }
    }

    if constexpr (EPI == VIDIL_EPI_F8) {
      // fp8 rows of 64 columns (64 B): [32][64] bytes, 16-B chunk index XOR ((row>>2)&3)
      {
        const int row = l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int cb = j * 32 + rq * 8 + hi * 4;
            *(uint32_t*)(ep + row * 64 + ((((cb >> 4) ^ ((row >> 2) & 3)) << 4) | (cb & 15))) =
                pack4_fp8(value(IT, j, rq, 0), value(IT, j, rq, 1), value(IT, j, rq, 2), value(IT, j, rq, 3));
          }
      }
      const int ch = lane & 3;
#pragma unroll
      for (int iter = 0; iter < 2; ++iter) {
        const int row = iter * 16 + (lane >> 2);
        const i32x4 v = *(const i32x4*)(ep + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4));
        const int m = m_w + IT * 32 + row;
        const int col = n_w + ch * 16;
        if (m < M && col + 16 <= N) *(i32x4*)((char*)p.out + (size_t)m * p.ldo + col) = v;
      }
    } else if constexpr (EPI == VIDIL_EPI_F16 || EPI == VIDIL_EPI_HEADS) {
      // 16-bit rows of 64 columns: [32][64] halfs, 16-B chunk index XOR (row&7)
      const float scale = (EPI == VIDIL_EPI_HEADS && part == 0) ? p.q_scale : 1.0f;
      {
        const int row = l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const f16x4 v = {Elt<TO>::from_f32(value(IT, j, rq, 0) * scale), Elt<TO>::from_f32(value(IT, j, rq, 1) * scale),
                             Elt<TO>::from_f32(value(IT, j, rq, 2) * scale), Elt<TO>::from_f32(value(IT, j, rq, 3) * scale)};
            *(f16x4*)(ep + row * 128 + (((j * 4 + rq) ^ (row & 7)) << 4) + hi * 8) = v;
          }
      }
      const int ch = lane & 7;
#pragma unroll
      for (int iter = 0; iter < 4; ++iter) {
        const int row = iter * 8 + (lane >> 3);
        const f16x8 v = *(const f16x8*)(ep + row * 128 + ((ch ^ (row & 7)) << 4));
        const int m = m_w + IT * 32 + row;
        const int col = n_w + ch * 8;
        if (m < M && col + 8 <= N) {
          if constexpr (EPI == VIDIL_EPI_F16) {
            *(f16x8*)((f16*)p.out + (size_t)m * p.ldo + col) = v;
          } else {
            const int b = m / p.T, t = m - b * p.T;
            const size_t bh = (size_t)b * p.H + head;
            if (part == 0) {
              *(f16x8*)((f16*)p.q + (bh * p.Tq_cap + t) * 64 + ch * 8) = v;
            } else {
              f16* kv = (f16*)(part == 1 ? p.k : p.vt);
              if (p.kv_tiled) {
                *(f16x8*)(kv + bh * p.Tk_cap * 64 + ktile_off(p.t_off + t, ch * 8)) = v;
              } else {
                *(f16x8*)(kv + (bh * p.Tk_cap + p.t_off + t) * 64 + ch * 8) = v;
              }
            }
          }
        }
      }
    } else {
      // f32 rows of 64 columns: [32][64] floats, chunk XOR (row&7); residual / position rows loaded before the LDS trip
      f32x4 rln_g = {1.f, 1.f, 1.f, 1.f}, rln_b = {0.f, 0.f, 0.f, 0.f};
      if constexpr (RLN) {
        const int col = n_w + (lane & 15) * 4;
        if (col + 4 <= N) {
          rln_g = *(const f32x4*)(p.rln_gamma + col);
          rln_b = *(const f32x4*)(p.rln_beta + col);
        }
      }
      f32x4 add[8];
      {
        const int ch = lane & 15;
        const int col = n_w + ch * 4;
#pragma unroll
        for (int iter = 0; iter < 8; ++iter) {
          const int lr = iter * 4 + (lane >> 4);
          const int m = m_w + IT * 32 + lr;
          add[iter] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (m < M && col + 4 <= N) {
            if constexpr (EPI == VIDIL_EPI_F32) {
              if (p.resid != nullptr) add[iter] = *(const f32x4*)(p.resid + (size_t)m * p.ldo + col);
            } else {  // EPI_PATCH
              const int t = m % p.tpi;
              add[iter] = *(const f32x4*)(p.pos + (size_t)(t + 1) * N + col);
            }
          }
          if constexpr (RLN) {
            const float rs = __shfl(st_s[IT], lr, 64), mrs = __shfl(st_ss[IT], lr, 64);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              add[iter][e] = __builtin_fmaf(__builtin_fmaf(add[iter][e], rs, -mrs), rln_g[e], rln_b[e]);
          }
        }
      }
      {
        const int lr = l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const f32x4 v = {value(IT, j, rq, 0), value(IT, j, rq, 1), value(IT, j, rq, 2), value(IT, j, rq, 3)};
            *(f32x4*)(ep + lr * 256 + (((j * 8 + rq * 2 + hi) ^ (lr & 7)) << 4)) = v;
          }
      }
      const int ch = lane & 15;
#pragma unroll
      for (int iter = 0; iter < 8; ++iter) {
        const int lr = iter * 4 + (lane >> 4);
        f32x4 v = *(const f32x4*)(ep + lr * 256 + ((ch ^ (lr & 7)) << 4));
        const int m = m_w + IT * 32 + lr;
        const int col = n_w + ch * 4;
        const bool ok = m < M && col + 4 <= N;
        if (ok) v += add[iter];
        if constexpr (EPI == VIDIL_EPI_F32 && STATS) {
          const f32x4 z = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
          const float s = row16_sum((z[0] + z[1]) + (z[2] + z[3]));
          const float ss = row16_sum((z[0] * z[0] + z[1] * z[1]) + (z[2] * z[2] + z[3] * z[3]));
          if (ch == 0 && m < M) *(f32x2*)(p.ln_stats_out + ((size_t)m * (N >> 6) + (n_w >> 6)) * 2) = f32x2{s, ss};
        }
        if (ok) {
          if constexpr (EPI == VIDIL_EPI_F32) {
            *(f32x4*)((float*)p.out + (size_t)m * p.ldo + col) = v;
            if (p.out16 != nullptr) {
              const f16x4 h4 = {Elt<TO>::from_f32(v[0]), Elt<TO>::from_f32(v[1]), Elt<TO>::from_f32(v[2]), Elt<TO>::from_f32(v[3])};
              *(f16x4*)((f16*)p.out16 + (size_t)m * p.ldo16 + col) = h4;
            }
          } else {  // EPI_PATCH
            const int b = m / p.tpi;
            *(f32x4*)((float*)p.out + ((size_t)m + b + 1) * p.ldo + col) = v;
          }
        }
      }
    }
  };

  // ================================================================================= issuer / epilogue phase
  // Runs beside the other group's main phase of tile q (q == nt: nobody computes any more — the last epilogue alone).
  // have_epi: this wave's accumulators hold tile q - 1.
  auto issuer_phase = [&](int q, bool have_epi) {
    const int s0 = q * nk;
    const bool live = q < nt;              // somebody consumes what is issued
    if (live) dma_setup(q);
    int dtile = q;                         // tile the DMA offsets describe
    if (have_epi) {
      int m0, n0;
      tile_origin(q - 1, m0, n0);
      m_w = m0 + wr * 128;
      n_w = n0 + wc * 64;
    }
    // Period i (between barriers B(s0+i) and B(s0+i+1); the main group computes element s0+i): the stage of element
    // s0+i-1 is free -> DMA of element s0+i+NST-1 into it; before the closing barrier element s0+i+1 must have landed:
    // it is the newest batch (2 stages: vmcnt(0)) or the batch before the newest (3 stages: vmcnt(12) leaves the 12 loads
    // issued last in flight; vmcnt retires in order, so everything older — epilogue stores included — is out).
    bool issued = false;
    auto issue = [&](int i) {
      issued = false;
      if (live) {
        const int j = i + NST - 1;         // K-tile of tile q, or (j - nk) of tile q + 1
        if (j < nk) {
          dma_issue(j, (s0 + j) % NST);
          issued = true;
        } else if (q + 1 < nt) {
          if (dtile != q + 1) {
            dma_setup(q + 1);
            dtile = q + 1;
          }
          dma_issue(j - nk, (s0 + j) % NST);
          issued = true;
        }
      }
    };
    auto close = [&]() {
      if (NST == 3 && issued) {
        asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    };
    // (2 stages: the DMA goes out first — it has only this period to land; 3 stages: after the piece, so that the
    //  piece's own loads and stores are older than it and the counted wait does not touch it)
#define VIDIL_PP_ITER(PI)                                             \
    if (NST == 2) issue(PI);                                         \
    if (have_epi) piece(std::integral_constant<int, PI>{});          \
    if (NST == 3) issue(PI);                                         \
    close();
    VIDIL_PP_ITER(0) VIDIL_PP_ITER(1) VIDIL_PP_ITER(2) VIDIL_PP_ITER(3)
    VIDIL_PP_ITER(4) VIDIL_PP_ITER(5) VIDIL_PP_ITER(6) VIDIL_PP_ITER(7)
#undef VIDIL_PP_ITER
    for (int i = 8; i < nk; ++i) {
      issue(i);
      close();
    }
  };

  // ================================================================================= the program of a wave
  if (grp == 1) {                          // the first issuer: elements 0 .. NST-2 of tile 0 (nk >= 8)
    dma_setup(0);
    dma_issue(0, 0);
    if (NST == 3) {
      dma_issue(1, 1);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __builtin_amdgcn_s_barrier();
  int pt = 0;
  if (grp == 1) {
    issuer_phase(0, false);
    pt = 1;
  }
  for (; pt < nt; pt += 2) {
    main_phase(pt);
    issuer_phase(pt + 1, true);
  }
}

template <typename T, int EPI, int ACT, bool FOLD = false, typename TO = T, bool STATS = false, bool RLN = false>
int launchpp(const vidil_gemm_args& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemmpp_kernel<T, TO, EPI, ACT, FOLD, STATS, RLN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds<EPI>);
    if (e != hipSuccess) {
      vidil_set_error("gemmpp: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
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
  const int ntiles = ((a.M + 255) / 256) * ((a.N + 127) / 128);
  const int grid = ntiles >= num_cu ? num_cu : (ntiles >= 8 ? (ntiles & ~7) : ntiles);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kLds<EPI>, s, a);
  VIDIL_CHECK_LAUNCH("gemmpp");
  return VIDIL_OK;
}

template <typename T>
int launchpp_dispatch(const vidil_gemm_args& a, hipStream_t s) {
  if (a.ln_fold) {
    if (a.epi == VIDIL_EPI_HEADS) return launchpp<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, true>(a, s);
    if (a.act == VIDIL_ACT_NONE) return launchpp<T, VIDIL_EPI_F16, VIDIL_ACT_NONE, true>(a, s);
    if (a.act == VIDIL_ACT_GELU_ERF) return launchpp<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF, true>(a, s);
    return launchpp<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU, true>(a, s);
  }
  switch (a.epi) {
    case VIDIL_EPI_F16:
      if (a.act == VIDIL_ACT_NONE) return launchpp<T, VIDIL_EPI_F16, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launchpp<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF>(a, s);
      return launchpp<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_F32:
      if (a.rln_gamma) return launchpp<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true, true>(a, s);
      if (a.ln_stats_out) return launchpp<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true>(a, s);
      if (a.act == VIDIL_ACT_NONE) return launchpp<T, VIDIL_EPI_F32, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launchpp<T, VIDIL_EPI_F32, VIDIL_ACT_GELU_ERF>(a, s);
      return launchpp<T, VIDIL_EPI_F32, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_HEADS:
      return launchpp<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE>(a, s);
    default:
      return launchpp<T, VIDIL_EPI_PATCH, VIDIL_ACT_NONE>(a, s);
  }
}

}  // namespace

// The ping-pong kernel serves a problem when it meets gemm256's alignment rules, has at least 8 K-tiles (the epilogue
// of a tile is spread over 8 of the next tile's K-tiles) and at least `min_tiles` 256x128 output tiles per CU-grid —
// below two tiles per workgroup there is nothing for an epilogue to hide under.  16-bit operands only for now.
bool vidil_gemmpp_eligible(const vidil_gemm_args& a) {
  static const int min_tiles = []() {
    const char* e = getenv("VIDIL_GEMMPP_MIN_TILES");
    return e ? atoi(e) : 512;
  }();
  if (const char* e = getenv("VIDIL_GEMMPP"))      // (read per call: tests and tools A/B the two kernels in one process)
    if (e[0] == '0') return false;
  if (a.dtype != VIDIL_DT_F16 && a.dtype != VIDIL_DT_BF16) return false;
  if (a.K / 64 < 8) return false;
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
  if (tiles < min_tiles) return false;
  return vidil_gemm256_eligible(a, true);
}

int vidil_gemmpp_launch(const vidil_gemm_args& a, hipStream_t s) {
  if (a.dtype == VIDIL_DT_BF16) return launchpp_dispatch<bf16>(a, s);
  return launchpp_dispatch<f16>(a, s);
}
