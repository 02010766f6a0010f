// gemm4w.hip — the large-M GEMM of the hot path, second generation: C[M,N] = A[M,K] · W[N,K]^T, 256x256x(128 B) tiles
// over FOUR waves (one per SIMD), each owning a 128x128 block of the tile in 256 accumulator registers.
//
// Why (round 3): with warm clocks the library GEMM beats the 8-wave kernel (gemm256.hip) by 7-15 % on the
// path's shapes and by 38 % at K = 8192, and its main loop is exactly this shape.  What the shape buys over 8 waves x (128x64):
//   * LDS fragment traffic per K-tile drops from 8 x 24 KiB to 4 x 32 KiB (a 128x128 register block reuses every
//     fragment 4 times instead of 2 / 4);
//   * one instruction stream per SIMD: the matrix pipe never arbitrates between two waves, and the schedule below is
//     written once, in program order: every MFMA is followed by at most one LDS read or one LDS-DMA piece;
//   * 2 barriers per K-tile (2,048 matrix-pipe cycles) instead of 4;
//   * the K-tile stream is CONTINUOUS across output tiles: the DMA head runs two K-tiles ahead of the MFMAs and walks
//     straight into the next output tile, so a tile's first K-tiles land during the previous tile's last iterations and
//     its epilogue (no per-tile prologue bubble; only the workgroup's very last two K-tiles are fetched twice).
//
// LDS (160 KiB, all of it): ring of 2 K-tiles x 4 half-tiles (A rows 0-127 / 128-255, W rows 0-127 / 128-255) of 16 KiB,
// layout as in gemm256.hip (128-B rows, 16-B slot XOR (row>>1)&7, filled by global_load_lds_dwordx4); then 4 x 8 KiB of
// private epilogue scratch (the ring is never idle here, so the transposition cannot borrow it).
//
// One iteration = one K-tile = 16*KS MFMAs per wave, fragments of the tile's first half of k-steps (S0) already in
// registers:
//     MFMAs of S0      | after each of the first ones: one fragment read of S1 (second half of k-steps, same tile)
//     ...              | lgkmcnt(0); BARRIER A: every wave has tile g in registers -> its ring buffer is free
//     ...              | 16 DMA pieces of tile g+2 into that buffer, one per 64 matrix-pipe cycles
//     MFMAs of S1      | ...
//     ...              | vmcnt(16); BARRIER B: tile g+1 (issued one iteration ago) is in LDS
//     ...              | fragment reads of S0 of tile g+1 (not in a tile's last iteration: the next output tile reads
//                      |   its first fragments after the epilogue — 64 registers less to carry across it)
// (A fragment register is overwritten only after the last MFMA that reads it has been issued; DMA of tile g+2 lands
// about 1.1 iterations ~ 1 us after it is issued — gemm256 gave it two half-periods, 0.4 us.)
//
// One wave per SIMD has nobody to cover its stalls, so the rest of the file is about not having any: no `s_waitcnt
// vmcnt(0)` while DMA or stores are in flight (which is what hipcc emits for a spill reload, for an ordinary load beside
// LDS-DMA, and at the end of an exec-masked block containing a load — see the comments at head_setup, issue_piece, the
// row partials and the lane ids), and instruction-level parallelism written into the program order of the epilogue
// (gemm_epilogue.inc under VIDIL_EPI_LAZY).  Results are bit-identical to gemm256's (tests/test_gemm4w_gpu.py).
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_epilogue.h"

#define VIDIL_EPI_LAZY 1   // gemm_epilogue.inc: bias / activation applied per stored quad, not to all accumulators up front
#define VIDIL_EPI_NIT TM   // ... over the TM 32-row tiles of a wave's block

namespace {

constexpr int SLOT = 16384;       // one half-tile: 128 rows x 128 B
// TM = 32-row tiles per wave: 4 -> 256 x 256 output tiles (A0, A1, W0, W1 per K-tile), 2 -> 128 x 256 tiles (A0, W0, W1:
// the form for problems with too few 256 x 256 tiles to fill the chip — 10,752 decode rows x 768 columns are 126 of
// those but 252 of these; the library picks the same tile there)
template <int TM> constexpr int kBuf = (TM == 4 ? 4 : 3) * SLOT;
template <int TM> constexpr int kRing = 2 * kBuf<TM>;
template <int TM> constexpr int kLds = kRing<TM> + 4 * 8192;

// compile-time loop: the body sees its index as a constant expression (accumulators and fragments are register arrays —
// every index into them must be static)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void glds16(const void* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// C3 (round 5): the ERROR-COMPENSATED product of the parity precision mode computed INSIDE the K loop.  A holds the rows
// [x_hi | x_lo | x_hi] (three planes of Kl = K / 3 columns: VIDIL_DT_SPLIT3's layout) and W the rows [W_hi | W_hi | W_lo], exactly
// as for the K-tripled plain launch — but instead of walking 3 Kl columns (every x_hi and every W_hi tile through LDS twice),
// one iteration fetches 32 logical columns of x_hi, x_lo, W_hi and W_lo — a 128-byte LDS row is [hi 64 B | lo 64 B], so the
// fragment loader's k-steps 0, 1 are the hi halves and 2, 3 the lo halves — and issues the three products from the same
// registers: (a_hi, w_hi), (a_hi, w_lo), (a_lo, w_hi) = 24 * TM MFMAs per 64 KiB of LDS-DMA where the plain loop issues 16 * TM:
// two thirds of the operand traffic (HBM, LDS-DMA and fragment reads alike) per MFMA.  The summation order differs from the
// K-tripled launch's (hi.hi + hi.lo + lo.hi per 32 columns instead of three passes over K), so a problem must take ONE of the two
// forms at every size: vidil_gemm routes split_k launches here whatever M is (TM = 2 for small grids: same k order, same bits).
template <typename T, typename TO, int EPI, int ACT, bool FOLD, bool STATS, bool RLN, int TM = 4, bool C3 = false>
__global__ __launch_bounds__(256) void gemm4w_kernel(const vidil_gemm_args p) {
  static_assert(TM == 4 || TM == 2, "256- or 128-row output tiles");
  static_assert(!C3 || (sizeof(T) == 2 && !FOLD && !RLN && !STATS), "the in-loop compensated product: 16-bit operands, plain epilogues");
  constexpr int ESZ_OF_T = sizeof(T);
  static_assert(TM == 4 || !(FOLD || RLN), "the 128-row form is built without the row-statistics epilogues");
  constexpr int BUF = kBuf<TM>, RING_BYTES = kRing<TM>;
  constexpr int MSH = TM == 4 ? 8 : 7;           // log2 of the tile's rows
  constexpr int NPIECE = TM == 4 ? 16 : 12;      // LDS-DMA instructions per K-tile per wave
  constexpr bool ROWSTAT = FOLD || RLN;
  // 16-bit row epilogues keep their transposition inside the lower 4 KiB of a wave's scratch: the upper 4 KiB receive the
  // bias / column-sum vectors by LDS-DMA during the last K-tile (gemm_epilogue.inc: VIDIL_EPI_BIAS_LDS)
  constexpr bool BIAS_LDS = EPI == VIDIL_EPI_F16 || EPI == VIDIL_EPI_HEADS || EPI == VIDIL_EPI_ARENA || EPI == VIDIL_EPI_F8;
  static_assert(!(FOLD && RLN), "a GEMM normalises either its A rows or its residual rows");
  static_assert(!RLN || EPI == VIDIL_EPI_F32, "the residual exists in the f32 epilogue only");
  using f16 = TO;
  using f16x4 = typename Elt<TO>::x4;
  using f16x8 = typename Elt<TO>::x8;
  using Frag = typename Mma<T>::Frag;
  constexpr int KS = Mma<T>::KS;
  constexpr int ESZ = sizeof(T);
  constexpr int KT = 128 / ESZ;
  static_assert(!FOLD || ESZ == 2, "the LayerNorm fold reads 16-bit A fragments");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 1;   // A half: output rows grp*128 ..
  const int wc2 = wave & 1;    // W half: output columns wc2*128 ..
  const int hi = lane >> 5, l31 = lane & 31;

  const int M = p.M, N = p.N, K = p.K;
  const int lda = p.lda > 0 ? p.lda : K;
  const int tiles_n = (N + 255) >> 8;
  const int tiles_m = (M + (1 << MSH) - 1) >> MSH;
  // developer experiment: a COLUMN-BLOCKED tile walk — blocks of `cblk` column tiles, all row panels
  // inside a block — so that the W rows an XCD works on at any one time stay inside its 4-MiB L2 (launch4w copies
  // $VIDIL_4W_COLBLOCK into the patch epilogue's `tpi` field for the other epilogues; 0 = the row-major walk)
  // (compiled in only with -DVIDIL_4W_COLBLOCK_EXP: the extra uniform state costs the hot instantiations 7-10 more spilled
  //  registers, which is not a price the shipped kernels pay for a developer knob)
#ifdef VIDIL_4W_COLBLOCK_EXP
  const int cblk = EPI == VIDIL_EPI_PATCH ? 0 : p.tpi;
#endif
  auto tile_of = [&](int lt, int& tm, int& tn) {
#ifdef VIDIL_4W_COLBLOCK_EXP
    if (cblk > 0) {
      const int per = tiles_m * cblk, cbk = lt / per, r = lt - cbk * per;
      tm = r / cblk;
      tn = cbk * cblk + (r - tm * cblk);
      return;
    }
#endif
    tm = lt / tiles_n;
    tn = lt - tm * tiles_n;
  };
  // persistent workgroups, XCD-contiguous tile ranges: as gemm256.hip
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
  constexpr int KBYTES = C3 ? 64 : 128;          // bytes of a plane's row one K-tile covers
  const int nk = C3 ? (K / 3) / 32 : K / KT;

  // ---- the DMA head: two K-tiles ahead of the MFMAs, walking this workgroup's tile sequence --------------------------
  // A thread's 16-B chunk of piece (half-tile hf, instruction i) is row hf*128 + i*32 + (tid >> 3) of the head tile, byte
  // column c16 of the K-tile.  Only the row (tid >> 3) and c16 live in registers; the offset of a piece is formed when it is
  // issued — min(row, last valid row) * row pitch + c16, three VALU operations beside 64 cycles of MFMA — on top of a
  // UNIFORM 64-bit base (the address form `global_load_lds_dwordx4 v_off, s[base:base+1]`).  Sixteen precomputed offsets
  // (the gemm256 way) were the registers this kernel could not afford: they spilled, and a spill reload in front of a DMA
  // piece is an `s_waitcnt vmcnt(0)` — a full drain of the stream.
  int h_r0, h_c16, h_c16w;
  {
    const int r0 = tid >> 3, sl = tid & 7;
    h_r0 = r0;
    h_c16 = (sl ^ ((r0 >> 1) & 7)) * 16;     // (rows 32 apart share the swizzle)
    h_c16w = h_c16;
    if constexpr (C3) {
      // source slot s of the LDS row: s < 4 -> 16-byte piece s of the hi plane's 64 bytes, s >= 4 -> piece s - 4 of the lo
      // plane's — plane 1 of an A row (Kl columns further), plane 2 of a W row (2 Kl columns further)
      const int ssl = sl ^ ((r0 >> 1) & 7);
      const int plane = (K / 3) * (int)sizeof(T);
      h_c16 = (ssl & 3) * 16 + (ssl >> 2) * plane;
      h_c16w = (ssl & 3) * 16 + (ssl >> 2) * 2 * plane;
    }
  }
  const char* hbaseA = (const char*)p.A;    // first row of the head tile's A panel / W panel
  const char* hbaseW = (const char*)p.W;
  int h_limA = 0, h_limW = 0;               // last valid row of the panel, relative to its first
  int h_logical = logical, h_remaining = remaining, h_kt = 0, h_buf = 0;
  bool h_live = true;    // false once the stream has run past the workgroup's last K-tile: the head then keeps re-fetching
  //                        that last K-tile into the free buffer (two wasted K-tiles per WORKGROUP) so that every
  //                        iteration issues its 16 pieces unconditionally and the counted waits stay uniform
  const uint32_t pitchA = (uint32_t)lda * ESZ, pitchW = (uint32_t)K * ESZ;
  auto head_setup = [&](int lt) {
    int tile_m, tile_n;
    tile_of(lt, tile_m, tile_n);
    const int hm0 = tile_m << MSH, hn0 = tile_n << 8;
    hbaseA = (const char*)p.A + (size_t)hm0 * pitchA;
    hbaseW = (const char*)p.W + (size_t)hn0 * pitchW;
    h_limA = M - 1 - hm0;
    h_limW = N - 1 - hn0;
  };
  head_setup(logical);
  // piece pc (0 .. NPIECE-1) of the head's K-tile: half-tiles W0, W1, A0 (, A1) in that order, 4 instructions of 4 KiB each;
  // LDS slots of a K-tile buffer: A0 (, A1), W0, W1
  constexpr int WSLOT = TM == 4 ? 2 : 1;
  auto issue_piece = [&](int pc) {
    const int sl = pc >> 2, i = pc & 3;
    char* dst = smem + h_buf * BUF + (sl < 2 ? WSLOT + sl : sl - 2) * SLOT + i * 4096 + wave * 1024;
    // (h_r0 made opaque at every piece: otherwise the compiler forms the eight row indices h_r0 + {0, 32, .. 224} once,
    //  keeps them across the tile loop and — when the epilogue needs the registers — spills them; their reloads then sit
    //  between the DMA pieces of an iteration, each one a VMEM load to be waited for)
    asm volatile("" : "+v"(h_r0));
    const int row = h_r0 + (sl & 1) * 128 + i * 32;
#if defined(VIDIL_4W_ABLATE) && (VIDIL_4W_ABLATE & 4)
    if (p.M > 0) return;     // developer ablation: no DMA (the MFMAs run on whatever LDS holds)
#endif
    if (sl < 2) {
      const int rr = row < h_limW ? row : h_limW;
      glds16(hbaseW + (size_t)(h_kt * KBYTES) + ((uint32_t)rr * pitchW + (uint32_t)(C3 ? h_c16w : h_c16)), dst);   // (< 2^32: vidil_gemm256_eligible)
    } else {
      const int rr = row < h_limA ? row : h_limA;
      glds16(hbaseA + (size_t)(h_kt * KBYTES) + ((uint32_t)rr * pitchA + (uint32_t)h_c16), dst);
    }
  };
  auto head_advance = [&]() {
    h_buf ^= 1;
    if (h_live && ++h_kt == nk) {
      if (h_remaining > tile_step) {
        h_kt = 0;
        h_logical += tile_step;
        h_remaining -= tile_step;
        head_setup(h_logical);
      } else {
        h_kt = nk - 1;
        h_live = false;
      }
    }
  };

  f32x16 accA[TM][2], accB[TM][2];   // output columns 0-63 / 64-127 of the wave's (32*TM) x 128 block
#define ACC(i, j) ((j) < 2 ? accA[i][(j) & 1] : accB[i][(j) & 1])
  float st_s[4], st_ss[4];
  f32x2 st_raw[4][4], st_sum[4];

  const int sw = (l31 >> 1) & 7;
  const int a_off = (TM == 4 ? grp * SLOT : grp * 8192) + l31 * 128;    // (TM == 2: rows grp*64 .. of the one A half-tile)
  const int w_off = (WSLOT + wc2) * SLOT + l31 * 128;
  Frag fa[KS][TM], fw[KS][4];
  constexpr int KH = KS / 2;           // k-steps per fragment set
  constexpr int NKS = 4 * TM;          // MFMAs per k-step per wave
  constexpr int NM = NKS * KS;         // MFMAs per K-tile per wave
  constexpr int NFK = 4 + TM;          // fragments per k-step: 4 W column tiles, TM A row tiles
  constexpr int NFR = KH * NFK;        // fragments per set
  constexpr int STEP = (ESZ == 2 && TM == 4) ? 2 : 1;   // MFMAs per DMA piece
#ifndef VIDIL_4W_TA
#define VIDIL_4W_TA 21
#endif
  constexpr int T_A = TM == 2 ? NFR + 1 : (ESZ == 2 ? VIDIL_4W_TA : 10);   // barrier A goes after this MFMA (developer sweep: -DVIDIL_4W_TA=17..29)
  constexpr int T_B = T_A + NPIECE * STEP;          // barrier B goes after this MFMA
  static_assert(T_B < NM - 2 && NFR <= T_A, "schedule");
  // ---- C3 schedule: six (A k-step, W k-step) products per K-tile; fragments sets S0 = k-steps 0, 1 (hi), S1 = 2, 3 (lo)
  //   pairs 0, 1: (0,0) (1,1)  hi.hi   | reads: fw[1] of THIS tile (deferred, see below), then S1 of this tile
  //   pairs 2, 3: (0,2) (1,3)  hi.lo   | barrier A, the DMA pieces of tile g+2, barrier B
  //   pairs 4, 5: (2,0) (3,1)  lo.hi   | S0 of tile g+1: the A fragments once pair 3 has issued (fa[0], fa[1] are dead),
  //                                    |   fw[0] once pair 4 has issued; fw[1] feeds pair 5 to the end of the iteration, so the next
  //                                    |   tile's fw[1] is read in the first slots of the next iteration (first needed by its pair 1)
  constexpr int NM3 = 6 * NKS;
  constexpr int C3_RS1 = 4;                                   // first slot of the S1 reads (after the deferred fw[1])
  constexpr int C3_TA = C3_RS1 + NFR + 1;                     // barrier A: every read of this tile has been issued
  constexpr int C3_STEP = TM == 4 ? 3 : 1;
  constexpr int C3_TB = C3_TA + NPIECE * C3_STEP;             // barrier B after this MFMA
  constexpr int C3_NA = (C3_TB + 2 > 4 * NKS ? C3_TB + 2 : 4 * NKS);      // first slot of the next tile's A fragments
  constexpr int C3_NW = (C3_NA + 2 * TM > 5 * NKS ? C3_NA + 2 * TM : 5 * NKS);   // first slot of the next tile's fw[0]
  static_assert(!C3 || (C3_NW + 4 <= NM3 && C3_TA + 1 <= 2 * NKS + NKS), "C3 schedule");
  auto read_frag = [&](const char* buf, int ks, int r) {     // r: 0-3 = W column tiles, 4-7 = A row tiles
#if defined(VIDIL_4W_ABLATE) && (VIDIL_4W_ABLATE & 8)
    if (p.M > 0) return;     // developer ablation: no fragment reads
#endif
    if (r < 4) fw[ks][r] = Mma<T>::load(buf + w_off + r * 4096, ks, hi, sw);
    else if (r - 4 < TM) fa[ks][r - 4] = Mma<T>::load(buf + a_off + (r - 4) * 4096, ks, hi, sw);
  };

  // ---- start of the stream: tiles 0 and 1, then S0 of tile 0 ------------------------------------------------------
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) issue_piece(pc);
  head_advance();
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) issue_piece(pc);
  head_advance();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
  __builtin_amdgcn_s_barrier();
  int cb = 0;   // ring buffer of the tile the MFMAs are on

  for (;;) {   // ======================================================================== one output tile
    int tile_m, tile_n;
    tile_of(logical, tile_m, tile_n);
    const int m0 = tile_m << MSH, n0 = tile_n << 8;
    // S0 of the tile's first K-tile (landed and visible since barrier B of the previous iteration / the start of the
    // stream).  Not fetched ahead across the epilogue: 64 live registers there cost more than this exposed LDS latency.
    static_for<NFR>([&](auto f_tag) {
      constexpr int f = decltype(f_tag)::value;
      read_frag(smem + cb * BUF, f / NFK, f % NFK);
    });
    auto iteration = [&](auto first_tag, auto last_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;   // first K-tile of an output tile: its first k-step starts from zero
      constexpr bool LAST = decltype(last_tag)::value;     // last K-tile: no fragment reads for a next one
      const char* const buf = smem + cb * BUF;
      const char* const nbuf = smem + (cb ^ 1) * BUF;
      static_for<NM>([&](auto n_tag) {
        constexpr int n = decltype(n_tag)::value;
        constexpr int ks = n / NKS, i = (n >> 2) % TM, j = n & 3;
        if constexpr (FIRST && ks == 0) {
          f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          // (fp8: C must be an OPAQUE zero tuple — with the constant 0 as C hipcc sent the result of every first-k-step
          //  v_mfma_scale_* through a[0:15] and scratch, 160 spilled registers; with an opaque C it initialises each accumulator
          //  tuple in place (16 v_accvgpr_write between two 64-cycle MFMAs) and accumulates: no spill at all)
          if constexpr (ESZ == 1) asm volatile("" : "+a"(zero));
          ACC(i, j) = Mma<T>::mma(fw[ks][j], fa[ks][i], zero);
        } else {
          ACC(i, j) = Mma<T>::mma(fw[ks][j], fa[ks][i], ACC(i, j));
        }
        if constexpr (n < NFR) read_frag(buf, KH + n / NFK, n % NFK);           // S1 of this tile
        if constexpr (n == T_A) {
          __builtin_amdgcn_s_waitcnt(0xC07F);                                    // lgkmcnt(0)
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (n > T_A && n <= T_B && (n - T_A - 1) % STEP == 0) {
          constexpr int pc = (n - T_A - 1) / STEP;
          issue_piece(pc);
          if constexpr (pc == NPIECE - 1) head_advance();
        }
        if constexpr (BIAS_LDS && LAST && n == 0) {
          // The epilogue's per-column vectors — bias, and the folded LayerNorm's column sums — for this wave's 128 columns,
          // fetched by ONE LDS-DMA instruction (lanes 0-31: bias, lanes 32-63: column sums, 16 B each) into the upper half
          // of the wave's epilogue scratch while the last K-tile is still being multiplied.  Loaded by the epilogue itself
          // (16 global loads per 64-column half) they cost an `s_waitcnt vmcnt(0)` — L2 latency, plus a drain of every store
          // of the previous half and of the DMA stream — twice per tile with the matrix pipe idle.  Older than this
          // iteration's DMA pieces, so barrier B's counted wait covers it (as for the row partials below).
          if (p.bias != nullptr || FOLD || ESZ_OF_T == 1) {     // (uniform)
            int lane_s;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
            int col = n0 + wc2 * 128 + (lane_s & 31) * 4;
            col = col + 4 <= N ? col : N - 4;      // (columns past N are never stored; N >= 4)
            const float* v0 = p.bias != nullptr ? p.bias : (FOLD ? p.ln_colsum : (ESZ_OF_T == 1 ? p.w_scale : (const float*)p.W));
            const float* v1 = FOLD ? p.ln_colsum : (ESZ_OF_T == 1 ? p.w_scale : v0);   // (fp8: the per-column weight scales)
            const float* src = (lane_s < 32 ? v0 : v1) + col;
            glds16(src, smem + RING_BYTES + wave * 8192 + 4096);
          }
        }
        if constexpr (ROWSTAT && LAST && n == 0) {
          // this half-wave's share of the producer's row partials (half-wave (wc2, hi) takes parts w, w+4, w+8, w+12 with
          // w = wc2 + 2*hi): issued at the top of the LAST K-tile (carried from the top of the output tile they were spilled
          // across the main loop), ahead of this iteration's DMA pieces, and summed behind barrier B below — where
          // "at most 16 operations in flight" already implies they have arrived, so no wait drains the DMA stream for them
          const int nparts = (FOLD ? K : N) >> 6;   // <= 16
          const f32x2* stats_in = (const f32x2*)p.ln_stats;
          int lane_s;   // (a fresh lane id: values derived from the kernel's own would be spilled across the main loop, and
          //               their reload HERE would be an `s_waitcnt vmcnt(0)` in the middle of the DMA stream)
          asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
          const int hi = lane_s >> 5, l31 = lane_s & 31;
          const int part0 = wc2 + 2 * hi;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            int row = m0 + grp * 128 + it * 32 + l31;
            row = row < M ? row : M - 1;
            // (straight-line loads, the part index clamped: parts past the end are masked where the sums are taken —
            //  a load in an exec-masked block would be waited for with vmcnt(0), draining the DMA stream mid-iteration)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int part = part0 + 4 * q;
              // (the load itself in inline asm: the compiler waits for a load it knows about with vmcnt(0) once LDS-DMA
              //  pieces are in flight beside it — it does not count across the two kinds — and that drains the stream;
              //  the wait that covers these loads is barrier B's vmcnt(16), they are older than its 16 pieces)
              const f32x2* src = stats_in + (size_t)row * nparts + (part < nparts ? part : nparts - 1);
              asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(st_raw[it][q]) : "v"(src));
            }
          }
        }
        if constexpr (n == T_B + 1) {
          // tile g+1 has landed once nothing older than this iteration's 16 DMA pieces is in flight
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (ROWSTAT && LAST && n == T_B + 2) {
          // the half-wave's sums, in gemm256's order: parts {w, w+8} then {w+4, w+12}; parts past the end count as zero
          int lane_s;
          asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
          const int nparts = (FOLD ? K : N) >> 6;
          const int part0 = wc2 + 2 * (lane_s >> 5);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              asm volatile("" : "+v"(st_raw[it][q]));   // (the loaded value exists from here on: behind barrier B's wait)
              if (part0 + 4 * q >= nparts) st_raw[it][q] = f32x2{0.f, 0.f};
            }
            st_sum[it] = f32x2{(st_raw[it][0][0] + st_raw[it][2][0]) + (st_raw[it][1][0] + st_raw[it][3][0]),
                               (st_raw[it][0][1] + st_raw[it][2][1]) + (st_raw[it][1][1] + st_raw[it][3][1])};
          }
        }
        if constexpr (n > T_B + 1 && !LAST) {                                    // S0 of the next K-tile
          constexpr int SLOTS = NM - T_B - 2;
          constexpr int PER = (NFR + SLOTS - 1) / SLOTS;
          constexpr int s = n - T_B - 2;
          static_for<PER>([&](auto q_tag) {
            constexpr int f = s * PER + decltype(q_tag)::value;
            if constexpr (f < NFR) read_frag(nbuf, f / NFK, f % NFK);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      cb ^= 1;
    };
    auto iteration_c3 = [&](auto first_tag, auto last_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr bool LAST = decltype(last_tag)::value;
      const char* const buf = smem + cb * BUF;
      const char* const nbuf = smem + (cb ^ 1) * BUF;
      static_for<NM3>([&](auto n_tag) {
        constexpr int n = decltype(n_tag)::value;
        constexpr int pr = n / NKS, i = (n >> 2) % TM, j = n & 3;
        constexpr int ka = pr == 0 ? 0 : pr == 1 ? 1 : pr == 2 ? 0 : pr == 3 ? 1 : pr == 4 ? 2 : 3;
        constexpr int kw = pr == 0 ? 0 : pr == 1 ? 1 : pr == 2 ? 2 : pr == 3 ? 3 : pr == 4 ? 0 : 1;
        if constexpr (FIRST && pr == 0) {
          f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          ACC(i, j) = Mma<T>::mma(fw[kw][j], fa[ka][i], zero);
        } else {
          ACC(i, j) = Mma<T>::mma(fw[kw][j], fa[ka][i], ACC(i, j));
        }
        if constexpr (!FIRST && n < 4) read_frag(buf, 1, n);                                  // fw[1] of this tile (deferred)
        if constexpr (n >= C3_RS1 && n < C3_RS1 + NFR) read_frag(buf, KH + (n - C3_RS1) / NFK, (n - C3_RS1) % NFK);   // S1
        if constexpr (n == C3_TA) {
          __builtin_amdgcn_s_waitcnt(0xC07F);                                    // lgkmcnt(0)
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (n > C3_TA && n <= C3_TB && (n - C3_TA - 1) % C3_STEP == 0) {
          constexpr int pc = (n - C3_TA - 1) / C3_STEP;
          issue_piece(pc);
          if constexpr (pc == NPIECE - 1) head_advance();
        }
        if constexpr (BIAS_LDS && LAST && n == 0) {
          if (p.bias != nullptr) {     // (uniform) the epilogue's bias vector for this wave's 128 columns by one LDS-DMA instruction
            int lane_s;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
            int col = n0 + wc2 * 128 + (lane_s & 31) * 4;
            col = col + 4 <= N ? col : N - 4;
            glds16(p.bias + col, smem + RING_BYTES + wave * 8192 + 4096);
          }
        }
        if constexpr (n == C3_TB + 1) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (!LAST) {                                                   // S0 of the next K-tile, except fw[1]
          if constexpr (n >= C3_NA && n < C3_NA + 2 * TM) read_frag(nbuf, (n - C3_NA) / TM, 4 + (n - C3_NA) % TM);   // fa[0][*], fa[1][*]
          if constexpr (n >= C3_NW && n < C3_NW + 4) read_frag(nbuf, 0, n - C3_NW);                                  // fw[0][*]
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      cb ^= 1;
    };
    if constexpr (C3) {
      if (nk == 1) {
        iteration_c3(std::true_type{}, std::true_type{});
      } else {
        iteration_c3(std::true_type{}, std::false_type{});
        for (int u = 2; u < nk; ++u) iteration_c3(std::false_type{}, std::false_type{});
        iteration_c3(std::false_type{}, std::true_type{});
      }
    } else if (nk == 1) {
      iteration(std::true_type{}, std::true_type{});
    } else {
      iteration(std::true_type{}, std::false_type{});
      for (int u = 2; u < nk; ++u) iteration(std::false_type{}, std::false_type{});
      iteration(std::false_type{}, std::true_type{});
    }

    // Everything after the main loop works from a FRESH lane id (volatile asm): its lane-dependent address arithmetic is
    // tile-invariant, and hoisted out of the tile loop it would sit in ~100 registers across the main loop and be spilled
    // — a spill reload between global stores (or right after the loop, with DMA in flight) costs an `s_waitcnt vmcnt(0)`,
    // i.e. a wait for every store / DMA piece issued so far (measured: 11 us per tile instead of 4).
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int lane = lane_e, hi = lane_e >> 5, l31 = lane_e & 31;
    if constexpr (ROWSTAT) {
      // row statistics, in gemm256's summation order (the two kernels agree bit for bit): half-wave (wc2, hi) plays
      // gemm256's wave w = wc2 + 2*hi — parts {w, w+8} then {w+4, w+12} — and the four per-"wave" sums meet in LDS and are
      // added in the order w = 0..3
      // (1-KiB blocks b = grp * 4 + w, two per wave's scratch, in its UPPER 4 KiB behind the bias / column-sum vectors that
      //  the last K-tile iteration fetched by LDS-DMA (VIDIL_EPI_BIAS_LDS: + 0 .. 1 KiB): the 16-bit epilogues of the FOLD
      //  kernels never write there, so no second barrier has to keep their transposition — lower 4 KiB — away from a late
      //  reader; the next tile's partials are written behind a main loop whose per-K-tile barriers every wave has to pass)
      auto stats_blk = [&](int b) { return (f32x2*)(smem + RING_BYTES + (b >> 1) * 8192 + 4096 + 1024 + (b & 1) * 1024); };
#pragma unroll
      for (int it = 0; it < 4; ++it) stats_blk(grp * 4 + wc2 + 2 * hi)[it * 32 + l31] = st_sum[it];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // (the divisor made opaque: 1 / K is tile-invariant, the compiler hoists it out of the tile loop, finds no register for
      //  it across the main loop and spills it — and its reload here is a VMEM load waited for with vmcnt(0): a drain of
      //  the DMA stream and of the previous tile's stores once per tile.  Recomputing the correctly rounded quotient costs
      //  ten instructions; the bits are those of gemm256's.)
      int kdiv = FOLD ? K : N;
      asm volatile("" : "+s"(kdiv));
      const float inv_k = 1.0f / (float)kdiv;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const f32x2 v = stats_blk(grp * 4 + w)[it * 32 + l31];
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
      if constexpr (RLN) {   // (the f32 epilogue transposes through ALL 8 KiB of a wave's scratch: late readers first)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }

    // ================================================================================ epilogue: two 128x64 halves
    const int m_w = m0 + grp * (32 * TM);
    char* const ep = smem + RING_BYTES + wave * 8192;
#if defined(VIDIL_4W_ABLATE) && (VIDIL_4W_ABLATE & 1)
    if (p.M > 0) {   // developer ablation: no epilogue (one dword per lane keeps the accumulators alive)
      float s = 0.f;
#pragma unroll
      for (int it = 0; it < TM; ++it)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += accA[it][j][0] + accB[it][j][5];
      if (s == 123.456f) ((float*)p.out)[lane] = s;
    } else
#endif
    {
#if defined(VIDIL_4W_ABLATE) && (VIDIL_4W_ABLATE & 2)
    const int M = p.M > 0 ? 0 : 1;    // developer ablation: the whole epilogue except its global stores (every row is "past M")
#endif
#define VIDIL_EPI_BIAS_LDS 1
    {
      const int n_w = n0 + wc2 * 128;
      const char* const epb = ep + 4096;            // [bias of columns n_w .. n_w+127 | column sums of the same]: 2 x 512 B
      do {
#define acc accA
#include "gemm_epilogue.inc"
#undef acc
      } while (0);
    }
    {
      const int n_w = n0 + wc2 * 128 + 64;
      const char* const epb = ep + 4096 + 256;
      do {
#define acc accB
#include "gemm_epilogue.inc"
#undef acc
      } while (0);
    }
    }
    if (remaining <= tile_step) break;
    logical += tile_step;
    remaining -= tile_step;
    // No drain of the epilogue's stores here: they share vmcnt with the LDS-DMA stream, and a counted wait with stores
    // still pending is merely conservative (loads complete in order among themselves, so "at most 16 operations
    // outstanding" still implies every DMA piece older than the newest 16 has landed; pending stores only make the
    // wait stricter).  The first MFMAs of the next tile run while this tile's stores are acknowledged.
  }
#undef ACC
}

template <typename T, int EPI, int ACT, bool FOLD = false, typename TO = T, bool STATS = false, bool RLN = false, int TM = 4, bool C3 = false>
int launch4w(const vidil_gemm_args& a, hipStream_t s) {
  static std::atomic<unsigned long long> attr_set{0};   // (one bit per device that has the opt-in: vidil_lds_opt_in)
  auto kern = gemm4w_kernel<T, TO, EPI, ACT, FOLD, STATS, RLN, TM, C3>;
  constexpr int LDS_BYTES = kLds<TM>;
  if (const int rc_ = vidil_lds_opt_in(attr_set, (const void*)kern, LDS_BYTES, "gemm4w")) return rc_;
  const int num_cu = vidil_cu_count() & ~7;     // (per device: core.hip)
  int cus = num_cu;
  if (const char* e = vidil_dev_env("VIDIL_GEMM_CUS")) {
    const int v = atoi(e) & ~7;
    if (v >= 8 && v < cus) cus = v;
  }
  const int ntiles = ((a.M + 64 * TM - 1) / (64 * TM)) * ((a.N + 255) / 256);
  // (fewer tiles than CUs: one workgroup per tile — the count rounded UP to the XCD multiple, the spare workgroups find
  //  their XCD's range empty and leave; rounded down, a few workgroups would run two tiles and double the launch's time)
  const int tiles = ntiles >= cus ? cus : (ntiles >= 8 ? ((ntiles + 7) & ~7) : ntiles);
#ifdef VIDIL_4W_COLBLOCK_EXP
  vidil_gemm_args a2 = a;
  if constexpr (EPI != VIDIL_EPI_PATCH) {
    a2.tpi = 0;
    if (const char* e = vidil_dev_env("VIDIL_4W_COLBLOCK")) {       // developer: column-blocked tile walk (see the kernel)
      const int cb = atoi(e), tn = (a.N + 255) / 256;
      if (cb > 0 && cb < tn && tn % cb == 0) a2.tpi = cb;
    }
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), LDS_BYTES, s, a2);
#else
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), LDS_BYTES, s, a);
#endif
  VIDIL_CHECK_LAUNCH("gemm4w");
  return VIDIL_OK;
}

template <typename T>
int launch4w_dispatch(const vidil_gemm_args& a, hipStream_t s) {
#ifdef VIDIL_4W_DEV_ONE   // developer builds (ISA inspection): one instantiation, VIDIL_4W_DEV_ONE = 1 fc1 (LN fold + GELU), 2 f32 + residual + row partials, 3 LN-folded heads, 4 = 2 + residual LayerNorm, 5 plain heads, 6 plain 16-bit
  if constexpr (VIDIL_4W_DEV_ONE == 1) return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF, true>(a, s);
  if constexpr (VIDIL_4W_DEV_ONE == 2) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true>(a, s);
  if constexpr (VIDIL_4W_DEV_ONE == 3) return launch4w<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, true>(a, s);
  if constexpr (VIDIL_4W_DEV_ONE == 4) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true, true>(a, s);
  if constexpr (VIDIL_4W_DEV_ONE == 5) return launch4w<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE>(a, s);
  if constexpr (VIDIL_4W_DEV_ONE == 6) return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_NONE>(a, s);
  if constexpr (VIDIL_4W_DEV_ONE == 7) return launch4w<fp8, VIDIL_EPI_F8, VIDIL_ACT_GELU_ERF, false, T>(a, s);     // (fp8 operands)
  if constexpr (VIDIL_4W_DEV_ONE == 8) return launch4w<fp8, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, false, T>(a, s);
  if constexpr (VIDIL_4W_DEV_ONE == 9) return launch4w<fp8, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T>(a, s);
  return -1000;
#else
  if (a.ln_fold) {
    if (a.epi == VIDIL_EPI_HEADS) return launch4w<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, true>(a, s);
    if (a.epi == VIDIL_EPI_ARENA) return launch4w<T, VIDIL_EPI_ARENA, VIDIL_ACT_NONE, true>(a, s);     // (round 6: the decode steps' Q|K|V)
    if (a.act == VIDIL_ACT_NONE) return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_NONE, true>(a, s);
    if (a.act == VIDIL_ACT_GELU_ERF) return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF, true>(a, s);
    return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU, true>(a, s);
  }
  switch (a.epi) {
    case VIDIL_EPI_F16:
      if (a.act == VIDIL_ACT_NONE) return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF>(a, s);
      return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_F32:
      if (a.rln_gamma) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true, true>(a, s);
      if (a.ln_stats_out) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, true>(a, s);
      if (a.act == VIDIL_ACT_NONE) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_NONE>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_GELU_ERF>(a, s);
      return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_QUICK_GELU>(a, s);
    case VIDIL_EPI_HEADS:
      return launch4w<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE>(a, s);
    case VIDIL_EPI_ARENA:
      return launch4w<T, VIDIL_EPI_ARENA, VIDIL_ACT_NONE>(a, s);
    default:
      return launch4w<T, VIDIL_EPI_PATCH, VIDIL_ACT_NONE>(a, s);
  }
#endif
}

}  // namespace

// the 128 x 256-tile form: plain epilogues only (the decode steps' projections and FFN, mid-size grids)
template <typename T>
int launch4w128_dispatch(const vidil_gemm_args& a, hipStream_t s) {
#ifdef VIDIL_4W_DEV_ONE
  return -1000;
#else
  if (a.ln_fold || a.ln_stats_out || a.rln_gamma) return -1000;
  switch (a.epi) {
    case VIDIL_EPI_F16:
      if (a.act == VIDIL_ACT_NONE) return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_NONE, false, T, false, false, 2>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_GELU_ERF, false, T, false, false, 2>(a, s);
      return launch4w<T, VIDIL_EPI_F16, VIDIL_ACT_QUICK_GELU, false, T, false, false, 2>(a, s);
    case VIDIL_EPI_F32:
      if (a.act == VIDIL_ACT_NONE) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, false, false, 2>(a, s);
      return -1000;
    case VIDIL_EPI_HEADS:
      return launch4w<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, false, T, false, false, 2>(a, s);
    case VIDIL_EPI_ARENA:
      return launch4w<T, VIDIL_EPI_ARENA, VIDIL_ACT_NONE, false, T, false, false, 2>(a, s);
    default:
      return -1000;
  }
#endif
}

// fp8 operands (round 4): the tower mode's GEMMs — fp8 hand-over with / without activation, per-head scatter into the 16-bit
// companion type, f32 + residual — on the 4-wave main loop (K-tiles of 128 e4m3, v_mfma_scale_f32_32x32x64_f8f6f4)
template <typename TO>
static int launch4w_fp8(const vidil_gemm_args& a, hipStream_t s) {
#ifdef VIDIL_4W_DEV_ONE
  return -1000;
#else
  if (a.ln_fold || a.ln_stats_out || a.rln_gamma || a.out16) return -1000;
  switch (a.epi) {
    case VIDIL_EPI_F32:
      if (a.act != VIDIL_ACT_NONE) return -1000;
      return launch4w<fp8, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, TO>(a, s);
    case VIDIL_EPI_HEADS: return launch4w<fp8, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, false, TO>(a, s);
    case VIDIL_EPI_F8:
      if (a.act == VIDIL_ACT_GELU_ERF) return launch4w<fp8, VIDIL_EPI_F8, VIDIL_ACT_GELU_ERF, false, TO>(a, s);
      if (a.act == VIDIL_ACT_QUICK_GELU) return launch4w<fp8, VIDIL_EPI_F8, VIDIL_ACT_QUICK_GELU, false, TO>(a, s);
      return launch4w<fp8, VIDIL_EPI_F8, VIDIL_ACT_NONE, false, TO>(a, s);
    default:
      return -1000;
  }
#endif
}

// split_k launches (the parity precision mode's GEMMs): the in-loop compensated product, at EVERY size — 256-row tiles once they
// fill the chip, 128-row tiles below (same k order per output element: same bits), so that a row's result never depends on the
// batch around it.  Epilogues the mode uses: f32 (+ activation, residual, the [hi | lo | hi] hand-over), the per-head scatter
// (cross K | V tiles; Q / K / V of the "16" attention kind), the patch embedding.
template <typename T, int TM>
static int launch4w_c3_tm(const vidil_gemm_args& a, hipStream_t s) {
  switch (a.epi) {
    case VIDIL_EPI_F32:
      if (a.act == VIDIL_ACT_NONE) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_NONE, false, T, false, false, TM, true>(a, s);
      if (a.act == VIDIL_ACT_GELU_ERF) return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_GELU_ERF, false, T, false, false, TM, true>(a, s);
      return launch4w<T, VIDIL_EPI_F32, VIDIL_ACT_QUICK_GELU, false, T, false, false, TM, true>(a, s);
    case VIDIL_EPI_HEADS:
      return launch4w<T, VIDIL_EPI_HEADS, VIDIL_ACT_NONE, false, T, false, false, TM, true>(a, s);
    case VIDIL_EPI_PATCH:
      if constexpr (TM == 4) return launch4w<T, VIDIL_EPI_PATCH, VIDIL_ACT_NONE, false, T, false, false, 4, true>(a, s);
      return -1000;
    default:
      return -1000;
  }
}
int vidil_gemm4w_c3_launch(const vidil_gemm_args& a, hipStream_t s) {
#ifdef VIDIL_4W_DEV_ONE
  return -1000;
#else
  const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  const bool big = t256 >= 5L * vidil_cu_count() / 8 || a.epi == VIDIL_EPI_PATCH;      // (the patch epilogue is built for 256-row tiles only)
  if (a.dtype == VIDIL_DT_BF16) return big ? launch4w_c3_tm<bf16, 4>(a, s) : launch4w_c3_tm<bf16, 2>(a, s);
  return big ? launch4w_c3_tm<f16, 4>(a, s) : launch4w_c3_tm<f16, 2>(a, s);
#endif
}

// tm: 4 = 256 x 256 tiles, 2 = 128 x 256 tiles.  -1000: the variant is not built here.
int vidil_gemm4w_launch(const vidil_gemm_args& a, hipStream_t s, int tm) {
  if (a.dtype == VIDIL_DT_FP8) return tm == 4 ? (a.dtype16 == VIDIL_DT_BF16 ? launch4w_fp8<bf16>(a, s) : launch4w_fp8<f16>(a, s)) : -1000;
  if (tm == 2) return a.dtype == VIDIL_DT_BF16 ? launch4w128_dispatch<bf16>(a, s) : launch4w128_dispatch<f16>(a, s);
  if (a.dtype == VIDIL_DT_BF16) return launch4w_dispatch<bf16>(a, s);
#ifdef VIDIL_4W_DEV_ONE
  return -1000;
#endif
  return launch4w_dispatch<f16>(a, s);
}
