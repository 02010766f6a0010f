// attention.hip — softmax(Q K^T) V for the short sequences on this path
// (ViT 197 keys, CLIP 50/77/257, MED text 1..35 queries over <=20 cached or 197
// image keys).
//
// Work unit = (key/value batch j, head h): all query batches that read j's K/V
// (uniform groups of `kv_group` consecutive query batches, or an explicit prefix
// table `group_start`) are flattened into "virtual rows" v = (qb - first)*Nq + t,
// so an image's K/V are fetched once for every caption / beam that attends to it.
//
// The kernels are built on v_mfma_f32_32x32x16_f16 with the scores computed
// TRANSPOSED (S^T = K·Q^T) so a lane owns one query row and the online softmax
// is lane-local (one cross-half shuffle per key tile), and with the output
// accumulated TRANSPOSED too (O^T = V^T·P^T) so the running rescale and the
// final 1/l are lane-local as well and a lane stores 4 contiguous d values:
//
//   attn_lds_kernel    rows > 32: K ([keys][64], rows padded to 144 B) and V^T
//                      ([64][keys], stride keys+8) staged once in LDS (strides
//                      chosen so the ds_read_b128 fragment reads are conflict
//                      free); each wave owns 32 virtual rows, 4 or 8 waves.
//   attn_direct1_kernel rows <= 32 (decode steps, prefill): every K/V element is
//                      used by one wave only, so fragments are loaded straight
//                      from HBM/L2 into MFMA operands (no LDS round trip); ONE wave
//                      owns a (batch, head) unit and walks its key tiles, the next
//                      tile requested before the current one is computed.
//   attn_direct_kernel the same with the 4 waves of a workgroup splitting a unit's
//                      key tiles (flash-decoding) and merging (m, l, O) through LDS:
//                      the form before the end of round 5, kept behind
//                      $VIDIL_ATTN_DIRECT1=0 for A / B measurements.
// (plus the streamed tower kernel, the one-tile wave kernel and, further down, the
//  f32 / split-operand kernels of the parity precision mode: each has its own header)
//
// The key order inside a 16-key MFMA step is the order the S^T accumulator
// layout leaves the P values in ({0-3,8-11} / {4-7,12-15} per half-wave).  V^T
// is STORED in that order (key t lives in column vt_pos(t), common.h: inside
// each 16-key block the 4-key groups 1 and 2 are swapped), so a half-wave's 8
// keys are one contiguous 16-byte read and P feeds the second MFMA without any
// permute.  NP (the V^T row stride) is a multiple of 16 for that reason.
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

template <typename T>
struct AttnP {
  const T* q;
  const T* k;
  const T* vt;
  T* out;
  const int32_t* kv_len;       // per query batch, or null
  const int32_t* kv_index;     // per query batch -> kv batch (only with kv_group == 1 semantics), or null
  const int32_t* group_start;  // [n_kv+1] prefix of query batches per kv batch, or null
  int Bq, H, Nq, Nk, Tq_cap, Tk_cap, NP, kv_group, causal, causal_off, ldo, n_kv;
  int out_mode;                // 0: T rows; 1: e4m3 bytes (the fp8 tower mode's proj-GEMM operand; store_rows kernels only);
                               // 2: error-compensated rows [hi | lo | hi] in three planes ldo/3 apart (VIDIL_DT_SPLIT3)
  int tiled;                   // K and V in 32-key fragment tiles (common.h: ktile_off / vtile_off); direct kernel only
  int rb;                      // staged kernel, single key chunk: rounds of NW row blocks per workgroup (launch_lds)
  int ostage;                  // staged kernel: 2 KiB of LDS per wave behind K / V^T for the output transposition (launch_lds:
                               // only where two workgroups per CU still fit with it)
  // direct kernel, split-operand form (vidil_attention_f32 arith = 1 with 16-bit K / V tiles): the queries are f32 rows read in
  // place — row (qb * Nq + t) * ldq32, head h at column h * 64 — scaled by q_scale and split into hi + lo in the kernel
  const float* q32;
  long long ldq32;
  float q_scale;
};

constexpr int KROW = 72;  // halfs per K row in LDS (64 + 8 pad)

struct RowInfo {
  int qb;     // query batch of this lane's virtual row
  int t;      // position inside the batch
  int klim;   // keys >= klim are excluded for this row
  bool valid;
};

// Resolve the work unit: kv batch + [first, first+count) query batches.
template <typename P>
__device__ __forceinline__ void resolve_unit(const P& p, int z, int& bk, int& first, int& count) {
  if (p.group_start != nullptr) {
    bk = z;
    first = p.group_start[z];
    count = p.group_start[z + 1] - first;
  } else if (p.kv_index != nullptr) {
    bk = p.kv_index[z];
    first = z;
    count = 1;
  } else {
    bk = z;
    first = z * p.kv_group;
    count = p.kv_group;
  }
}

template <typename P>
__device__ __forceinline__ RowInfo row_info(const P& p, int v, int first, int rows) {
  RowInfo r;
  r.valid = v < rows;
  const int vc = r.valid ? v : rows - 1;
  const int g = vc / p.Nq;
  r.qb = first + g;
  r.t = vc - g * p.Nq;
  int klim = p.Nk;
  if (p.kv_len != nullptr) {
    const int L = p.kv_len[r.qb];
    klim = L < klim ? L : klim;
  }
  if (p.causal) {
    const int c = r.t + p.causal_off + 1;
    klim = c < klim ? c : klim;
  }
  r.klim = klim;
  return r;
}

// One key tile of online softmax + P·V for the 32 rows of a wave.
// S: scores of this tile (S^T layout: lane = row, 16 keys per half-wave).
// MASKED = false: the caller knows (wave-uniformly) that every key of the tile is inside every row's limit — written as
// a run-time `need_mask` alone the compiler turns the 16 tests into selects that run on every tile (49 of a tile's ~160
// VALU instructions in the towers' self-attention, whose 197 keys need the mask on the last tile only).
//
// The online softmax keeps a REFERENCE value m per row, not the exact running maximum: m moves (and l and O are
// rescaled) only when a tile's maximum outgrows it by more than kLazy = 5 — decided for the whole wave by one ballot, so
// on most tiles the 32 accumulator multiplies, the exp of the correction and the dependent chain through them are not
// executed at all.  Any reference is exact in exact arithmetic (softmax is shift invariant); a stale one lets the
// probabilities of a tile reach e^5 = 148 instead of 1, far inside the range of the 16-bit operand they are rounded to,
// and the row sum stays in f32.  p = 2^(s*log2e - m*log2e): one FMA and one v_exp_f32 per score.
#ifndef VIDIL_ATTN_KLAZY
#define VIDIL_ATTN_KLAZY 5.0f
#endif
constexpr float kLazy = VIDIL_ATTN_KLAZY;   // (developer: -DVIDIL_ATTN_KLAZY=0.0f is the exact running maximum)
constexpr float kLog2e = 1.44269504088896340736f;
// Part 1 of a key tile: scores S -> probabilities rounded to T and packed as the two 16-key operands of P.V (pf), with the
// row's reference m, its sum l and — when the reference moves — the accumulators O brought up to date.
// FIRST = true (a row block's first tile, the caller's promise): nothing is accumulated yet, so m starts at the tile's
// maximum and nothing is rescaled; the general form computes exactly the same values there (alpha = 0 on zeros).
// LAZY = false: the reference is the exact running maximum and the rescale is unconditional — straight-line code for the
// direct kernel, which is HBM-bound and lives on having every load of a wave in flight before its first MFMA: with the
// ballot's branch in the tile the compiler drains the loads before it (decode cross-attention 396 -> 493 us per launch).
template <typename T, bool MASKED = true, bool FIRST = false, bool LAZY = true>
__device__ __forceinline__ void softmax_tile(f32x16& S, int key0, int klim, bool need_mask, float& m, float& l, f32x16 (&O)[2],
                                             typename Elt<T>::x8 (&pf)[2]) {
  const int hi = (threadIdx.x & 63) >> 5;
  if (MASKED && need_mask) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= klim) S[r] = -INFINITY;
    }
  }
  float mt = S[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mt = fmaxf(mt, S[r]);
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
  if (FIRST) {
    m = mt;
  } else if (!LAZY) {
    const float mn = fmaxf(m, mt);
    const float alpha = __builtin_amdgcn_exp2f((m - (mn == -INFINITY ? 0.f : mn)) * kLog2e);   // m == -inf -> 0
    l *= alpha;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
  } else {
    const bool grow = mt > m + kLazy;            // (m == -inf: true as soon as the row has seen one finite score)
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
      const float mn = grow ? mt : m;            // the two lanes of a row decide alike (mt is shared above)
      // rows that keep their reference: alpha = 2^0 = 1 exactly; m == -inf (nothing accumulated yet): alpha = 0
      const float alpha = __builtin_amdgcn_exp2f((m - (mn == -INFINITY ? 0.f : mn)) * kLog2e);
      l *= alpha;
      m = mn;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
    }
  }
  const float mc = (m == -INFINITY ? 0.f : m) * kLog2e;
  float ps = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], kLog2e, -mc));
    S[r] = e;
    ps += e;
  }
  l = FIRST ? ps : l + ps;
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[hb][j] = (T)S[hb * 8 + j];
}

// One key tile of online softmax + P·V for the 32 rows of a wave.
// S: scores of this tile (S^T layout: lane = row, 16 keys per half-wave).
template <typename T, bool MASKED = true, bool LAZY = true, bool FIRST = false, typename VFrag>
__device__ __forceinline__ void softmax_pv_tile(f32x16& S, int key0, int klim, bool need_mask, float& m, float& l,
                                                f32x16 (&O)[2], VFrag&& vfrag) {
  typename Elt<T>::x8 pf[2];
  softmax_tile<T, MASKED, FIRST, LAZY>(S, key0, klim, need_mask, m, l, O, pf);
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const typename Elt<T>::x8 vf = vfrag(dt, hb);
      // O^T[d][q] += V^T[d][keys] · P^T[keys][q]
      O[dt] = Elt<T>::mfma32(vf, pf[hb], O[dt]);
    }
  }
}

// The same tile with the PROBABILITIES as hi + lo (split-operand form: O += V.P_lo + V.P_hi, the values 16-bit as stored): exact
// running maximum, straight-line code.
template <typename T, typename VFrag>
__device__ __forceinline__ void softmax_pv_tile_psplit(f32x16& S, int key0, int klim, bool need_mask, float& m, float& l,
                                                       f32x16 (&O)[2], VFrag&& vfrag) {
  const int hi = (threadIdx.x & 63) >> 5;
  if (need_mask) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= klim) S[r] = -INFINITY;
    }
  }
  float mt = S[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mt = fmaxf(mt, S[r]);
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
  const float mn = fmaxf(m, mt);
  const float mc = (mn == -INFINITY ? 0.f : mn) * kLog2e;
  const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m, kLog2e, -mc));   // m == -inf -> 0
  l *= alpha;
  m = mn;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
  typename Elt<T>::x8 ph[2], pl[2];
  float ps = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], kLog2e, -mc));
    ps += e;
    const T eh = Elt<T>::from_f32(e);
    ph[r >> 3][r & 7] = eh;
    pl[r >> 3][r & 7] = Elt<T>::from_f32(e - (float)eh);
  }
  l += ps;
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const typename Elt<T>::x8 vf = vfrag(dt, hb);
      O[dt] = Elt<T>::mfma32(vf, pl[hb], O[dt]);
      O[dt] = Elt<T>::mfma32(vf, ph[hb], O[dt]);
    }
  }
}

// O[dt][r] is (d = dt*32 + (r&3) + 8*(r>>2) + 4*hi, row = lane&31): 4 contiguous d per register quad.
template <typename T>
__device__ __forceinline__ void store_rows(const AttnP<T>& p, const RowInfo& ri, int h, const f32x16 (&O)[2], float inv) {
  using f16x4 = typename Elt<T>::x4;
  using f16 = T;
  if (!ri.valid) return;
  const int hi = (threadIdx.x & 63) >> 5;
  if (p.out_mode >= 2) {   // wave-uniform: hi = T(o), lo = T(o - hi), planes [hi | lo | hi] (3: hi | lo only)
    const int pl = p.ldo / 3;
    f16* og = p.out + ((size_t)ri.qb * p.Nq + ri.t) * p.ldo + h * 64 + 4 * hi;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f16x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = O[dt][rq * 4 + e] * inv;
          vh[e] = Elt<T>::from_f32(v);
          vl[e] = Elt<T>::from_f32(v - (float)vh[e]);
        }
        *(f16x4*)(og + dt * 32 + rq * 8) = vh;
        *(f16x4*)(og + pl + dt * 32 + rq * 8) = vl;
        if (p.out_mode == 2) *(f16x4*)(og + 2 * pl + dt * 32 + rq * 8) = vh;
      }
    return;
  }
  if (p.out_mode == 1) {   // wave-uniform
    uint8_t* o8 = (uint8_t*)p.out + ((size_t)ri.qb * p.Nq + ri.t) * p.ldo + h * 64 + 4 * hi;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *(uint32_t*)(o8 + dt * 32 + rq * 8) = pack4_fp8(O[dt][rq * 4 + 0] * inv, O[dt][rq * 4 + 1] * inv, O[dt][rq * 4 + 2] * inv,
                                                        O[dt][rq * 4 + 3] * inv);
    return;
  }
  f16* og = p.out + ((size_t)ri.qb * p.Nq + ri.t) * p.ldo + h * 64 + 4 * hi;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f16x4 v = {(f16)(O[dt][rq * 4 + 0] * inv), (f16)(O[dt][rq * 4 + 1] * inv), (f16)(O[dt][rq * 4 + 2] * inv),
                       (f16)(O[dt][rq * 4 + 3] * inv)};
      *(f16x4*)(og + dt * 32 + rq * 8) = v;
    }
}

// The 16-bit rows of a wave's 32 x 64 output block, TRANSPOSED through 2 KiB of wave-private LDS so that the global
// stores are 16 B per lane and 64 contiguous bytes per row (store_rows above writes 8 B per lane with consecutive lanes
// 1.5 KB apart: 512 partial-sector writes per block — compiled out, the staged tower kernel ran 18 % faster).  Two
// passes of 32 d each; the 16-byte chunk index is XORed with (row >> 1) & 3 (no bank conflicts either way).  Rows of a
// block may belong to different query batches: a row's destination offset and validity come from the lane that owns it.
template <typename T>
__device__ __forceinline__ void store_rows_lds(const AttnP<T>& p, const RowInfo& ri, int h, const f32x16 (&O)[2], float inv, char* scratch) {
  using f16 = T;
  using f16x4 = typename Elt<T>::x4;
  using f16x8 = typename Elt<T>::x8;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 5, l31 = lane & 31;
  // element offset of this lane's row (valid rows only) — fetched below by the lanes that store that row
  const long long own = ri.valid ? (long long)(((size_t)ri.qb * p.Nq + ri.t) * p.ldo + h * 64) : -1;
  long long dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = i * 16 + (lane >> 2);
    const int lo = __shfl((int)(own & 0xffffffffLL), r, 64), hi32 = __shfl((int)(own >> 32), r, 64);
    dst[i] = ((long long)hi32 << 32) | (unsigned int)lo;
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f16x4 v = {(f16)(O[dt][rq * 4 + 0] * inv), (f16)(O[dt][rq * 4 + 1] * inv), (f16)(O[dt][rq * 4 + 2] * inv),
                       (f16)(O[dt][rq * 4 + 3] * inv)};
      *(f16x4*)(scratch + l31 * 64 + ((rq ^ ((l31 >> 1) & 3)) << 4) + hi * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = i * 16 + (lane >> 2), c = lane & 3;
      const f16x8 v = *(const f16x8*)(scratch + r * 64 + ((c ^ ((r >> 1) & 3)) << 4));
      if (dst[i] >= 0) *(f16x8*)(p.out + dst[i] + dt * 32 + c * 8) = v;
    }
  }
}

// ------------------------------------------------------------------ LDS-staged kernel
template <typename T, int NKT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) void attn_lds_kernel(const AttnP<T> p) {
  using f16 = T;                       // (the body below is written in terms of "the 16-bit operand type")
  using f16x8 = typename Elt<T>::x8;
  constexpr int NKEY = NKT * 32;
  constexpr int VROW = NKEY + 8;  // halfs per V^T row in LDS: 16-B aligned rows, (2*NKEY+16)/16 odd ->
                                  // the ds_read_b128 of 16 different rows hit 16 different 16-B bank slots
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* Ks = (f16*)smem;                      // [NKEY][KROW]
  f16* Vs = (f16*)(smem + NKEY * KROW * 2);  // [64][VROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  int bk, first, count;
  resolve_unit(p, blockIdx.z, bk, first, count);
  const int rows = count * p.Nq;
  const int nk = p.Nk;
  // Rows of a unit per workgroup: NW blocks of 32, or — when all keys fit one staged chunk — `rb` rounds of them
  // (wave w then walks blocks w, w+NW, ...): a unit whose rows spill a little over NW*32 (8 captions x 35 tokens
  // = 280 ITM rows) is then served by ONE staging of its K/V instead of two.
  const int rows_per_wg = NW * 32 * (nk <= NKEY ? p.rb : 1);
  const int base = blockIdx.x * rows_per_wg;
  if (base >= rows) return;  // uniform: this row tile is empty for this unit

  const f16* kg = p.k + ((size_t)bk * p.H + h) * p.Tk_cap * 64;
  const f16* vg = p.vt + ((size_t)bk * p.H + h) * 64 * (size_t)p.NP;
  // Staging issues EVERY global load of a thread before the first LDS store (compile-time trip counts, predicated):
  // written as load -> store loops with run-time bounds the compiler kept one load in flight per thread, and a
  // workgroup spent 4-7 serial HBM round trips here before its first MFMA.
  constexpr int KIT = (NKEY * 8 + NT - 1) / NT;          // 16-byte chunks of K per thread
  constexpr int VIT = (64 * NKT * 4 + NT - 1) / NT;       // 16-byte chunks of V^T per thread (same count: NKEY * 8)
  auto stage = [&](int k0) {
    f16x8 kreg[KIT], vreg[VIT];        // (both batches of loads are issued before the first store below)
    // ---- K rows k0 .. k0+NKEY-1 (rows >= Nk zero) -------------------------------------------------------
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      const int q = tid + i * NT;
      const int row = q >> 3, c = q & 7;
      kreg[i] = zero8<T>();
      if (q < NKEY * 8 && k0 + row < nk) kreg[i] = *(const f16x8*)(kg + (size_t)(k0 + row) * 64 + c * 8);
    }
    if (p.NP == 0) {
      // ---- V given ROW-MAJOR ([keys][64], NP == 0: the QKV GEMM then stores V like K, 16 B per lane, instead of
      //      scattering V^T with 2-byte stores): transpose while staging.  Lanes walk consecutive keys, so for a
      //      fixed d the 64 lanes of a wave write 128 contiguous bytes of one V^T row (no bank conflicts); the
      //      16-byte global reads of a wave cover 64 key rows and are re-used from L1 by the next d-chunk.
      const f16* vrow = p.vt + ((size_t)bk * p.H + h) * p.Tk_cap * 64;
#pragma unroll
      for (int i = 0; i < KIT; ++i) {
        const int q = tid + i * NT;
        const int c = q / NKEY, r = q - c * NKEY;
        vreg[i] = zero8<T>();
        if (q < NKEY * 8 && k0 + r < nk) vreg[i] = *(const f16x8*)(vrow + (size_t)(k0 + r) * 64 + c * 8);
      }
    } else {
      // ---- V^T columns k0 .. (keys >= Nk zero: masked P is 0 but 0*garbage must stay 0) -------------------
#pragma unroll
      for (int i = 0; i < VIT; ++i) {
        const int q = tid + i * NT;
        const int d = q / (NKT * 4), kc = q - d * (NKT * 4);
        const int pos0 = k0 + kc * 8;           // storage columns pos0..pos0+7 (key = vt_pos(column))
        vreg[i] = zero8<T>();
        if (q < 64 * NKT * 4 && (pos0 & ~15) < nk && pos0 + 8 <= p.NP) vreg[i] = *(const f16x8*)(vg + (size_t)d * p.NP + pos0);
      }
    }
    // ---- LDS stores ------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      const int q = tid + i * NT;
      if (q < NKEY * 8) *(f16x8*)(Ks + (q >> 3) * KROW + (q & 7) * 8) = kreg[i];
    }
    if (p.NP == 0) {
#pragma unroll
      for (int i = 0; i < KIT; ++i) {
        const int q = tid + i * NT;
        const int c = q / NKEY, r = q - c * NKEY;
        if (q < NKEY * 8) {
          const int col = vt_pos(r);
#pragma unroll
          for (int e = 0; e < 8; ++e) Vs[(c * 8 + e) * VROW + col] = vreg[i][e];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < VIT; ++i) {
        const int q = tid + i * NT;
        const int d = q / (NKT * 4), kc = q - d * (NKT * 4);
        const int pos0 = k0 + kc * 8;
        if (q < 64 * NKT * 4) {
          f16x8 v = vreg[i];
          if ((pos0 | 15) + 1 > nk) {            // the 16-key block that straddles Nk: zero the keys past the end
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (vt_pos(pos0 + e) >= nk) v[e] = (f16)0.f;
          }
          *(f16x8*)(Vs + d * VROW + kc * 8) = v;
        }
      }
    }
  };

  auto stage_next = [&](int k0) {   // chunks after the first (sequences over NKEY keys): the accumulators and Q
    // fragments are live here, so this one keeps a single load in flight per thread instead of a batch
    // ---- stage K rows k0 .. k0+NKEY-1 (rows >= Nk zero) ------------------------------------------------
    for (int q = tid; q < NKEY * 8; q += NT) {
      const int row = q >> 3, c = q & 7;
      f16x8 v = zero8<T>();
      if (k0 + row < nk) v = *(const f16x8*)(kg + (size_t)(k0 + row) * 64 + c * 8);
      *(f16x8*)(Ks + row * KROW + c * 8) = v;
    }
    // ---- V given ROW-MAJOR ([keys][64], NP == 0: the QKV GEMM then stores V like K, 16 B per lane, instead of
    //      scattering V^T with 2-byte stores): transpose while staging.  Lanes walk consecutive keys, so for a
    //      fixed d the 64 lanes of a wave write 128 contiguous bytes of one V^T row (no bank conflicts); the
    //      16-byte global reads of a wave cover 64 key rows and are re-used from L1 by the next d-chunk.
    if (p.NP == 0) {
      const f16* vrow = p.vt + ((size_t)bk * p.H + h) * p.Tk_cap * 64;
      for (int q = tid; q < NKEY * 8; q += NT) {
        const int c = q / NKEY, r = q - c * NKEY;
        f16x8 v = zero8<T>();
        if (k0 + r < nk) v = *(const f16x8*)(vrow + (size_t)(k0 + r) * 64 + c * 8);
        const int col = vt_pos(r);
#pragma unroll
        for (int e = 0; e < 8; ++e) Vs[(c * 8 + e) * VROW + col] = v[e];
      }
    }
    // ---- stage V^T columns k0 .. (keys >= Nk zero: masked P is 0 but 0*garbage must stay 0) --------------
    for (int q = tid; p.NP != 0 && q < 64 * NKT * 4; q += NT) {
      const int d = q / (NKT * 4), kc = q - d * (NKT * 4);
      const int pos0 = k0 + kc * 8;           // storage columns pos0..pos0+7 (key = vt_pos(column))
      const int blk_end = (pos0 | 15) + 1;    // end of the 16-key block this chunk belongs to
      f16x8 v = zero8<T>();
      if (blk_end <= nk) {
        v = *(const f16x8*)(vg + (size_t)d * p.NP + pos0);
      } else if ((pos0 & ~15) < nk && pos0 + 8 <= p.NP) {
        v = *(const f16x8*)(vg + (size_t)d * p.NP + pos0);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (vt_pos(pos0 + e) >= nk) v[e] = (f16)0.f;
      }
      *(f16x8*)(Vs + d * VROW + kc * 8) = v;
    }
  };

  // Keys are consumed in chunks of NKEY (one chunk when Nk <= NKEY; 577-key ViT@384 sequences take three
  // 224-key chunks): stage the chunk, run its key tiles through the online softmax, re-stage.  Multi-chunk
  // launches keep one row block per wave (its (m, l, O) lives across the chunks).
  const int nblk = nk <= NKEY ? p.rb : 1;
  // The first row block's Q fragments and the first K/V chunk are requested together, before anything else is live.
  RowInfo ri;
  f16x8 qf[4];
  bool active;
  auto load_q = [&](int rbi) {
    const int v0 = base + (rbi * NW + wave) * 32;
    active = v0 < rows;                        // waves without rows still stage and meet the barriers
    ri = row_info(p, (active ? v0 : 0) + l31, first, rows);
    const f16* qg = p.q + (((size_t)ri.qb * p.H + h) * p.Tq_cap + ri.t) * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qg + ks * 16);
  };
  load_q(0);
  stage(0);
  __syncthreads();
#pragma unroll 1
  for (int rbi = 0; rbi < nblk; ++rbi) {
    if (rbi > 0) {
      load_q(rbi);
      if (!active) break;                      // (single chunk: nothing left to stage, no barrier ahead)
    }
    // smallest klim in the wave decides from which tile on masking is needed
    int kmin = ri.klim;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int x = __shfl_xor(kmin, o, 64);
      kmin = x < kmin ? x : kmin;
    }

    float m = -INFINITY, l = 0.f;
    f32x16 O[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;

#pragma unroll 1
    for (int k0 = 0; k0 < nk; k0 += NKEY) {
      if (rbi == 0 && k0 > 0) {
        __syncthreads();                        // every wave is done reading the previous chunk
        stage_next(k0);
        __syncthreads();
      }
      if (!active) continue;

#pragma unroll 1
      for (int kt = 0; kt < NKT; ++kt) {
        if (k0 + kt * 32 >= nk) break;
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const f16x8 kf = *(const f16x8*)(Ks + (kt * 32 + l31) * KROW + ks * 16 + hi * 8);
          S = Elt<T>::mfma32(kf, qf[ks], S);
        }
        const bool need_mask = (k0 + kt * 32 + 32) > kmin;
        auto vfrag = [&](int dt, int hb) { return *(const f16x8*)(Vs + (dt * 32 + l31) * VROW + (kt * 2 + hb) * 16 + 8 * hi); };
        // (a scalar branch to the form without the 16 key tests: as one run-time flag they become selects on every tile)
        if (need_mask) softmax_pv_tile<T, true>(S, k0 + kt * 32, ri.klim, true, m, l, O, vfrag);
        else softmax_pv_tile<T, false>(S, k0 + kt * 32, ri.klim, false, m, l, O, vfrag);
      }
    }
    if (!active) break;
    l += __shfl_xor(l, 32, 64);
    if (p.out_mode == 0 && p.ostage)   // (uniform) plain 16-bit rows: through the wave's LDS scratch, 16-B / 64-B-per-row stores
      store_rows_lds(p, ri, h, O, l > 0.f ? 1.0f / l : 0.f, smem + NKEY * KROW * 2 + 64 * VROW * 2 + wave * 2048);
    else
      store_rows(p, ri, h, O, l > 0.f ? 1.0f / l : 0.f);
  }
}

// ------------------------------------------------------------------ streamed kernel (tower self-attention)
// The towers' self-attention (197 keys, V row-major, one query batch per K/V batch, no masks) is the one big launch of
// the staged kernel above, and there it waits: global -> registers -> LDS staging, a barrier, the key tiles, the stores —
// phases of ONE unit that only a second workgroup on the CU overlaps.  This form keeps a workgroup on the CU for many
// units (unit = (image, head), walked with the grid's stride) and streams their K / V through LDS by LDS-DMA
// (`global_load_lds_dwordx4`: no staging registers, so the seven computing waves stay at four waves per SIMD):
//
//   wave 7 (has no query rows at <= 224 rows) is the producer: a unit's NKT key tiles live in NKT fixed slots (K tile
//     [32 keys][64] + V tile [32 keys][64]).  The consumers read a K tile one step before its V tile, so the halves of a
//     unit are "K tiles 0..KA-1 + V tiles 0..KA-2" and the rest: the producer requests the second half of the current unit
//     right after barrier B1 and the first half of the NEXT unit right after barrier B2, and arrives at each barrier only
//     after `vmcnt(0)` — it is the only wave that ever waits for the stream.
//   waves 0-6 own 32 query rows each.  Per unit: B1, K.Q^T of tile 0, then per key tile the softmax (VALU) followed by
//     ONE block of eight MFMAs — P.V of this tile and K.Q^T of the next, as alternating independent chains — B2 before the
//     block that first touches the second half; the next unit's Q fragments are requested half a unit ahead (into a
//     second register set: under load a request takes 3-4 us), the output rows go out transposed through the wave's 2 KiB
//     of LDS by buffer stores.  Two barriers per unit; between them the waves drift, which is what lets one wave's MFMAs
//     run under another's softmax.
//
// LDS images are what the fragment reads want, chosen through the SOURCE address each DMA lane fetches (the LDS side of
// a DMA instruction is lane-linear):
//   K tile: 128-B rows, 16-B slot c of row r at slot c ^ ((r >> 1) & 7) (the GEMMs' swizzle; ds_read_b128, conflict free)
//   V tile: [d / 16][32 keys][16 d] (32-B rows; the four d-blocks 1152 B apart, so the two a half-wave reads together
//     sit 32 banks apart): the P.V operand — 8 keys of one d per lane — is read TRANSPOSED by two `ds_read_b64_tr_b16`
//     (each 16-lane group reads one [4 keys][16 d] block, 128 contiguous bytes; tools/micro/tr16_probe.hip pins the
//     instruction's lane mapping).  Row-major V needs no key permutation: the S^T accumulator leaves a half-wave the
//     keys {0-3, 8-11} / {4-7, 12-15} of a 16-key step, which are whole 4-key blocks.
// Rows past the last key are fetched from the last key's row (finite values; their probabilities are exactly 0).
// The arithmetic — MFMA order per accumulator, online softmax, key -> k-slot assignment — is the staged kernel's:
// outputs are bit-identical (tests/test_kernels_gpu.py).  Measured (MI355X, 3,584 images x 12 heads, bf16, same process):
// 1,090-1,100 us staged -> 900-920 us; `SQ_VALU_MFMA_BUSY` 0.26 -> 0.34 of the kernel's cycles.  What the ablations say
// (tools/bench_attn_stream.py on -DVIDIL_ATTN_ABLATE builds): arithmetic alone 670 us, + DMA 780-800, + stores 780, all
// 900-920; LDS bank conflicts are gone (0.07 of the LDS cycles, all in the output transposition), the split point KA, the
// Q prefetch distance and s_setprio around the MFMA block each move it by < 2 %.
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void attn_glds16(const void* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// store_rows_lds for the streamed kernel: the same transposition, but every lane takes part in exactly four 16-byte BUFFER
// stores per block (rows past the unit's last one get an out-of-range offset and are dropped by the range check) — no
// exec-masked branch around a store, so the number of memory operations between the next unit's Q loads and their first
// use is a constant and the wait for Q does not become a wait for the stores' acknowledgements.
// `row_off`: byte offset of this lane's own row inside the unit's output window, or 0x80000000 for a row that does not exist.
template <typename T>
__device__ __forceinline__ void store_rows_stream(__amdgpu_buffer_rsrc_t rsrc, uint32_t row_off, const f32x16 (&O)[2], float inv,
                                                  char* scratch) {
  using f16 = T;
  using f16x4 = typename Elt<T>::x4;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 5, l31 = lane & 31;
  uint32_t dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) dst[i] = (uint32_t)__shfl((int)row_off, i * 16 + (lane >> 2), 64);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f16x4 v = {(f16)(O[dt][rq * 4 + 0] * inv), (f16)(O[dt][rq * 4 + 1] * inv), (f16)(O[dt][rq * 4 + 2] * inv),
                       (f16)(O[dt][rq * 4 + 3] * inv)};
      *(f16x4*)(scratch + l31 * 64 + ((rq ^ ((l31 >> 1) & 3)) << 4) + hi * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = i * 16 + (lane >> 2), c = lane & 3;
      const u32x4 v = *(const u32x4*)(scratch + r * 64 + ((c ^ ((r >> 1) & 3)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, dst[i] + (uint32_t)((dt * 32 + c * 8) * 2), 0, 0);
    }
  }
}

// The fp8 tower mode's rows (e4m3 bytes, the proj GEMM's operand): 32 rows x 64 bytes through the same 2 KiB, two 16-byte
// buffer stores per lane.  `row_off` is in bytes of the fp8 row (ldo counts bytes there).
__device__ __forceinline__ void store_rows_stream_fp8(__amdgpu_buffer_rsrc_t rsrc, uint32_t row_off, const f32x16 (&O)[2], float inv,
                                                      char* scratch) {
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 5, l31 = lane & 31;
  uint32_t dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) dst[i] = (uint32_t)__shfl((int)row_off, i * 16 + (lane >> 2), 64);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      // bytes dt*32 + rq*8 + hi*4 .. +3 of the row; 16-byte chunk index XORed with (row >> 1) & 3 as in the 16-bit form
      const int chunk = dt * 2 + (rq >> 1), in_chunk = (rq & 1) * 8 + hi * 4;
      *(uint32_t*)(scratch + l31 * 64 + ((chunk ^ ((l31 >> 1) & 3)) << 4) + in_chunk) =
          pack4_fp8(O[dt][rq * 4 + 0] * inv, O[dt][rq * 4 + 1] * inv, O[dt][rq * 4 + 2] * inv, O[dt][rq * 4 + 3] * inv);
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = i * 16 + (lane >> 2), c = lane & 3;
    const u32x4 v = *(const u32x4*)(scratch + r * 64 + ((c ^ ((r >> 1) & 3)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, dst[i] + (uint32_t)(c * 16), 0, 0);
  }
}

template <typename T, int NKT, bool OUT8>   // OUT8: rows of e4m3 bytes (the fp8 tower mode) instead of 16-bit rows
__global__ __launch_bounds__(512, 4) void attn_stream_kernel(const AttnP<T> p) {
  using f16x8 = typename Elt<T>::x8;
#ifndef VIDIL_ATTN_KA
#define VIDIL_ATTN_KA ((NKT + 1) / 2)
#endif
  constexpr int KA = VIDIL_ATTN_KA;      // key tiles of the first half of a unit (developer sweep: -DVIDIL_ATTN_KA=2..NKT-1)
  constexpr int VSUB = 1152;             // bytes between the four [32 keys][16 d] blocks of a V tile: 1 KiB + 128, so that
                                         // the two blocks a half-wave reads together (d 0-15 / 16-31) sit 32 banks apart
  constexpr int TILE = 4096 + 4 * VSUB;  // K tile 4 KiB + V tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nunits = p.n_kv * p.H;
  const int nk = p.Nk;
  const size_t kv_unit = (size_t)p.Tk_cap * 128;     // bytes of one unit's K (or V)
  const int stride = gridDim.x;
  int u = blockIdx.x;
  if (u >= nunits) return;

  if (wave == 7) {
    // ------------------------------------------------------------------ producer
    // piece pc of a tile (1 KiB, one DMA instruction): pc 0-3 K rows pc*8 .. pc*8+7, pc 4-7 V columns (pc-4)*16 .. +15
    auto issue_half = [&](int uu, int kt, int pc0) {     // pc0 = 0: the K tile, 4: the V tile
      const char* gb = (const char*)(pc0 == 0 ? p.k : p.vt) + (size_t)uu * kv_unit;
      char* dst = smem + kt * TILE;
#pragma unroll
      for (int pc = pc0; pc < pc0 + 4; ++pc) {
        int r, cb;
        if (pc < 4) {
          r = pc * 8 + (lane >> 3);
          cb = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        } else {
          r = lane >> 1;
          cb = (pc - 4) * 32 + (lane & 1) * 16;
        }
        int key = kt * 32 + r;
        key = key < nk - 1 ? key : nk - 1;
#if defined(VIDIL_ATTN_ABLATE) && (VIDIL_ATTN_ABLATE & 1)
        if (uu != (int)blockIdx.x) continue;     // developer ablation: DMA for a workgroup's first unit only
#endif
        attn_glds16(gb + (uint32_t)(key * 128 + cb), dst + (pc < 4 ? pc * 1024 : 4096 + (pc - 4) * VSUB));
      }
    };
    // A K tile is read one step before its V tile (the consumers' pipeline), so the two halves of a slot are free and
    // needed at different barriers: "first half" = K tiles 0..KA-1 and V tiles 0..KA-2, "second half" = the rest.
    auto issue_first = [&](int uu) {
#pragma unroll
      for (int kt = 0; kt < KA; ++kt) issue_half(uu, kt, 0);
#pragma unroll
      for (int kt = 0; kt < KA - 1; ++kt) issue_half(uu, kt, 4);
    };
    auto issue_second = [&](int uu) {
#pragma unroll
      for (int kt = KA; kt < NKT; ++kt) issue_half(uu, kt, 0);
#pragma unroll
      for (int kt = KA - 1; kt < NKT; ++kt) issue_half(uu, kt, 4);
    };
    issue_first(u);
    for (;;) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // B1: first half of u in LDS; every wave is done with unit u - stride
      issue_second(u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // B2: second half of u in LDS; K tiles < KA, V tiles < KA-1 of u are done with
      u += stride;
      if (u >= nunits) break;
      issue_first(u);
    }
    return;
  }

  // -------------------------------------------------------------------- consumers
  const int hi = lane >> 5, l31 = lane & 31;
  const int sw = (l31 >> 1) & 7;
  const int row = wave * 32 + l31;
  const bool valid = row < p.Nq;
  const int vc = valid ? row : p.Nq - 1;
  // transposed V reads: lane t of 16-lane group g supplies row (t >> 2), 8-byte chunk (t & 3) of its group's block
  const int vlane = ((lane >> 4) & 1) * VSUB + (hi * 4 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
  char* scratch = smem + NKT * TILE + wave * 2048;
  f16x8 qf[4];
  const uint32_t q_off = (uint32_t)(vc * 128 + hi * 16);     // (one register: the unit's base is uniform)
  // (buffer loads: descriptor + scalar unit offset + one lane register — as a 64-bit lane address the compiler keeps a
  //  register pair per wave for it and, at 128 registers, spills it)
  const __amdgpu_buffer_rsrc_t rsrc_q = uniform_rsrc(p.q, (uint32_t)nunits * p.Tq_cap * 128);
  auto load_q = [&](int uu, f16x8 (&dst)[4]) {
    const uint32_t ub = (uint32_t)uu * p.Tq_cap * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      dst[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, q_off + ks * 32, ub, 0));
  };
  const uint32_t row_off = valid ? (uint32_t)row * p.ldo * (OUT8 ? 1 : 2) : 0x80000000u;   // bytes inside the unit's output window
  load_q(u, qf);
  asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]));   // (arrived: the loop is entered with nothing in flight)
  // Software pipeline of a row block: the four MFMAs of P.V of tile kt and the four of the NEXT tile's K.Q^T are issued
  // as one block of alternating, mutually independent chains (O[0], S, O[1], S, ...) right after tile kt's probabilities
  // are packed — the matrix pipe gets eight back-to-back instructions instead of two dependent chains with a softmax
  // between them, and tile kt+1's scores are ready when the wave's VALU work on them starts.
  // fragment reads of one HALF (hb) of a step: V^T of tile kv (both d tiles, keys hb*16 .. +15) and K of tile kk (k-steps
  // 2*hb, 2*hb+1) — 16 registers; the two halves of a step reuse them (all eight fragments at once do not fit beside the
  // accumulators, the scores, two sets of Q and the packed probabilities in 128 registers)
  auto read_k2 = [&](int kt, int h2, f16x8 (&kf)[2]) {
    const char* ks_ = smem + kt * TILE + l31 * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) kf[j] = *(const f16x8*)(ks_ + ((((h2 * 2 + j) * 2 + hi) ^ sw) << 4));
  };
  auto read_v2 = [&](int kt, int hb, f16x8 (&vf)[2]) {     // [dt]
    const char* vs_ = smem + kt * TILE + 4096 + vlane + hb * 512;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const char* a = vs_ + dt * (2 * VSUB);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 256));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
      vf[dt] = __builtin_bit_cast(f16x8, both);
    }
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifndef VIDIL_ATTN_QAHEAD
#define VIDIL_ATTN_QAHEAD 2
#endif
  constexpr int QAHEAD = VIDIL_ATTN_QAHEAD;               // (developer sweep: 1 .. NKT-2)
  f16x8 qn[4];
#ifdef VIDIL_ATTN_TIMING   // developer build: cycles a wave spends at the two barriers / in the store phase / in total -> out
  uint64_t t_b1 = 0, t_b2 = 0, t_st = 0;
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    float m, l;
    f32x16 O[2], S;
    f16x8 pf[2];
#ifdef VIDIL_ATTN_TIMING
    const uint64_t tb1 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();                         // B1: K tiles 0..KA-1 / V tiles 0..KA-2 of this unit are in LDS
    asm volatile("" ::: "memory");
#ifdef VIDIL_ATTN_TIMING
    t_b1 += __builtin_amdgcn_s_memtime() - tb1;
#endif
    // one step: P.V of tile kt (probabilities pf) fused with K.Q^T of tile kt + 1, in two halves of four MFMAs
    auto fused = [&](int kt, auto first_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
      f16x8 kf[2], vf[2];
#ifdef VIDIL_ATTN_PRIO
      __builtin_amdgcn_s_setprio(VIDIL_ATTN_PRIO);
#endif
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        read_v2(kt, hb, vf);
        read_k2(kt + 1, hb, kf);
        O[0] = Elt<T>::mfma32(vf[0], pf[hb], FIRST && hb == 0 ? zero16 : O[0]);
        S = Elt<T>::mfma32(kf[0], qf[hb * 2], hb == 0 ? zero16 : S);
        O[1] = Elt<T>::mfma32(vf[1], pf[hb], FIRST && hb == 0 ? zero16 : O[1]);
        S = Elt<T>::mfma32(kf[1], qf[hb * 2 + 1], S);
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef VIDIL_ATTN_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    };
    {
      f16x8 kf[2];
      read_k2(0, 0, kf);
      S = Elt<T>::mfma32(kf[0], qf[0], zero16);
      S = Elt<T>::mfma32(kf[1], qf[1], S);
      read_k2(0, 1, kf);
      S = Elt<T>::mfma32(kf[0], qf[2], S);
      S = Elt<T>::mfma32(kf[1], qf[3], S);
    }
    // ---- tile 0: nothing accumulated yet
    softmax_tile<T, false, true>(S, 0, nk, false, m, l, O, pf);
    fused(0, std::true_type{});
    // ---- tiles 1 .. NKT-2 (all keys inside the limit)
#pragma unroll 1
    for (int kt = 1; kt < NKT - 1; ++kt) {
      // (uniform) the next unit's Q, requested a good half unit ahead: under load a request takes 3-4 us to come back — the
      // time of two or three key tiles — and a wave that waits for its Q at a unit's start holds up the barrier for all.
      // The last workgroups' last pass re-reads their own unit instead of branching: the operation count stays the same.
      if (kt == QAHEAD) load_q(u + stride < nunits ? u + stride : u, qn);
      softmax_tile<T, false>(S, kt * 32, nk, false, m, l, O, pf);
#ifdef VIDIL_ATTN_TIMING
      const uint64_t tb2 = __builtin_amdgcn_s_memtime();
#endif
      if (kt == KA - 1) __builtin_amdgcn_s_barrier();     // B2: K tiles KA.. / V tiles KA-1.. are in LDS
      asm volatile("" ::: "memory");
#ifdef VIDIL_ATTN_TIMING
      if (kt == KA - 1) t_b2 += __builtin_amdgcn_s_memtime() - tb2;
#endif
      fused(kt, std::false_type{});
    }
    const int un = u + stride;
    const int bk = u / p.H, h = u - bk * p.H;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];       // (tile NKT-2 has just released qf)
    // ---- last tile: the one that may straddle the key count
    softmax_tile<T, true>(S, (NKT - 1) * 32, nk, true, m, l, O, pf);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      f16x8 vf[2];
      read_v2(NKT - 1, hb, vf);
      O[0] = Elt<T>::mfma32(vf[0], pf[hb], O[0]);
      O[1] = Elt<T>::mfma32(vf[1], pf[hb], O[1]);
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
#ifdef VIDIL_ATTN_TIMING
    const uint64_t ts0 = __builtin_amdgcn_s_memtime();
#endif
#if defined(VIDIL_ATTN_ABLATE) && (VIDIL_ATTN_ABLATE & 2)
    if (inv == 12345.f)                           // developer ablation: no output stores
#endif
    {
      // the unit's output window: rows bk*Nq .. +Nq-1, this head's 64 columns onwards (out_mode 1: rows of e4m3 bytes)
      if constexpr (OUT8) {
        const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc((const char*)p.out + ((size_t)bk * p.Nq * p.ldo + h * 64), (uint32_t)p.Nq * p.ldo);
        store_rows_stream_fp8(rsrc, row_off, O, inv, scratch);
      } else {
        const __amdgpu_buffer_rsrc_t rsrc = uniform_rsrc(p.out + ((size_t)bk * p.Nq * p.ldo + h * 64), (uint32_t)p.Nq * p.ldo * 2);
        store_rows_stream<T>(rsrc, row_off, O, inv, scratch);
      }
    }
#ifdef VIDIL_ATTN_TIMING
    t_st += __builtin_amdgcn_s_memtime() - ts0;
#endif
    if (un >= nunits) break;
    u = un;
  }
#ifdef VIDIL_ATTN_TIMING
  if (lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t* dbg = (uint64_t*)p.out + ((size_t)blockIdx.x * 8 + wave) * 4;
    dbg[0] = __builtin_amdgcn_s_memtime() - t_begin;
    dbg[1] = t_b1;
    dbg[2] = t_b2;
    dbg[3] = t_st;
  }
#endif
}

// ------------------------------------------------------------------ direct (no K/V staging) kernel
// rows <= 32.  4 waves; wave w handles key tiles w, w+4, ...; partials merged by wave 0.
// QS (split-operand form, the parity precision mode's decode cross-attention): f32 queries split into hi + lo in the kernel and
// the probabilities split alike — S = K.Q_lo + K.Q_hi, O += V.P_lo + V.P_hi — against the SAME 16-bit K / V tiles.  Which of the
// four operands need more than 16 bits was measured on the CPU oracle (tests/probes/probe_precision_design.py, trained-like
// statistics, of the logit scale): Q 1.7e-4, P 6.3e-5, V 2.1e-5, K 1.0e-5 — with Q and P split the 16-bit K / V leave 3.0e-5 where
// the budget of "1e-3 absolute" at max|logit| = 16 is 6.4e-5, at the byte traffic of the plain kernel (f32 K / V: twice the bytes).
template <typename T, int NKT, bool QS = false, int VAR = 0>   // VAR 2 (the QS form at 197 keys): three waves per SIMD, one key tile per wave and round
__global__ __launch_bounds__(256, VAR ? 3 : 1) void attn_direct_kernel(const AttnP<T> p) {
  using f16 = T;
  using f16x8 = typename Elt<T>::x8;
  __shared__ float part_m[4][32];
  __shared__ float part_l[4][32];
  __shared__ float part_o[4][64][33];  // [wave][d][row] (+1 pad)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  int bk, first, count;
  resolve_unit(p, blockIdx.z, bk, first, count);
  const int rows = count * p.Nq;
  if (rows <= 0) return;  // uniform: a kv batch nobody attends to
  const int nk = p.Nk;
  const RowInfo ri = row_info(p, l31, first, rows);

  float m = -INFINITY, l = 0.f;
  f32x16 O[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;

  const f16* kg = p.k + ((size_t)bk * p.H + h) * p.Tk_cap * 64;
  const f16* vg = p.vt + ((size_t)bk * p.H + h) * 64 * (size_t)(p.tiled ? p.Tk_cap : p.NP);
  const int ntiles = (nk + 31) >> 5;
  if (wave < ntiles) {
    f16x8 qf[4], ql[QS ? 4 : 1];
    if constexpr (QS) {
      const float* qg = p.q32 + ((size_t)ri.qb * p.Nq + ri.t) * p.ldq32 + h * 64 + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x4 a = *(const f32x4*)(qg + ks * 16) * p.q_scale, b = *(const f32x4*)(qg + ks * 16 + 4) * p.q_scale;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qf[ks][e] = Elt<T>::from_f32(a[e]);
          ql[ks][e] = Elt<T>::from_f32(a[e] - (float)qf[ks][e]);
          qf[ks][4 + e] = Elt<T>::from_f32(b[e]);
          ql[ks][4 + e] = Elt<T>::from_f32(b[e] - (float)qf[ks][4 + e]);
        }
      }
    } else {
      const f16* qg = p.q + (((size_t)ri.qb * p.H + h) * p.Tq_cap + ri.t) * 64 + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qg + ks * 16);
    }
    int kmin = ri.klim;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int x = __shfl_xor(kmin, o, 64);
      kmin = x < kmin ? x : kmin;
    }
    // This kernel streams each K / V^T byte exactly once and is HBM-bound, so every load of the wave's (up
    // to NI) key tiles is issued before the first MFMA: 16 x 16 B per lane in flight instead of 4.
    // Sequences longer than 8 key tiles (577 image tokens at 384^2) go through rounds of 2 tiles per wave.
    constexpr int NI = VAR == 2 ? 1 : (NKT >= 8 ? 2 : (NKT + 3) / 4);
    constexpr int ROUNDS = (NKT + 4 * NI - 1) / (4 * NI);
#pragma unroll 1
    for (int round = 0; round < ROUNDS; ++round) {
    const int kt0 = round * 4 * NI + wave;
    if (kt0 >= ntiles) break;
    f16x8 kf[NI][4], vf[NI][2][2];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int kt = kt0 + 4 * i;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[i][ks] = zero8<T>();
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) vf[i][dt][hb] = zero8<T>();
      if (kt < ntiles && p.tiled) {
        // fragment tiles: each instruction of the wave reads one contiguous KiB; rows / key groups past the
        // last key are not fetched (they stay zero and are masked below)
        const f16* kt_base = kg + (size_t)kt * 2048 + (hi * 32 + l31) * 8;
        if (kt * 32 + l31 < nk) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) kf[i][ks] = *(const f16x8*)(kt_base + ks * 512);
        }
        const f16* vt_base = vg + (size_t)kt * 2048 + (hi * 32 + l31) * 8;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          if ((kt * 2 + hb) * 16 + 4 * hi < nk) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vf[i][dt][hb] = *(const f16x8*)(vt_base + (hb * 2 + dt) * 512);
          }
        }
      } else if (kt < ntiles) {
        int krow = kt * 32 + l31;
        krow = krow < nk ? krow : nk - 1;  // clamped rows are masked below
        const f16* kr = kg + (size_t)krow * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[i][ks] = *(const f16x8*)(kr + ks * 16);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          // this half-wave's 8 keys of the 16-key block are 16 contiguous bytes (vt_pos order)
          const int blk0 = (kt * 2 + hb) * 16;
          if (blk0 < nk) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) vf[i][dt][hb] = *(const f16x8*)(vg + (size_t)(dt * 32 + l31) * p.NP + blk0 + 8 * hi);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int kt = kt0 + 4 * i;
      if (kt >= ntiles) break;
      f32x16 S;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = 0.f;
      if constexpr (QS) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) S = Elt<T>::mfma32(kf[i][ks], ql[ks], S);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) S = Elt<T>::mfma32(kf[i][ks], qf[ks], S);
      const bool tail = (kt * 32 + 32) > nk;  // tile reaches past the last key: V^T needs zeroing too
      const bool need_mask = (kt * 32 + 32) > kmin;
      auto vfrag = [&](int dt, int hb) {
        const int blk0 = (kt * 2 + hb) * 16;
        f16x8 v = vf[i][dt][hb];
        if (tail) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (blk0 + 4 * hi + (e & 3) + 8 * (e >> 2) >= nk) v[e] = (f16)0.f;
        }
        return v;
      };
      // (a wave's first tile, known at compile time when one round covers the keys: nothing to rescale)
      if constexpr (QS) softmax_pv_tile_psplit<T>(S, kt * 32, ri.klim, need_mask, m, l, O, vfrag);
      else if (ROUNDS == 1 && i == 0) softmax_pv_tile<T, true, false, true>(S, kt * 32, ri.klim, need_mask, m, l, O, vfrag);
      else softmax_pv_tile<T, true, false>(S, kt * 32, ri.klim, need_mask, m, l, O, vfrag);
    }
    }  // rounds
  }
  // ---- merge the waves' partials --------------------------------------------------------------
  l += __shfl_xor(l, 32, 64);
  if (hi == 0) { part_m[wave][l31] = m; part_l[wave][l31] = l; }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) part_o[wave][dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi][l31] = O[dt][r];
  __syncthreads();
  // all 256 threads: thread = (row = tid&31, d-block = tid>>5 of 8 d values)
  {
    const int row = tid & 31, db = tid >> 5;
    if (row < rows) {
      float mm = part_m[0][row];
#pragma unroll
      for (int w = 1; w < 4; ++w) mm = fmaxf(mm, part_m[w][row]);
      float sc[4], lt = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        sc[w] = part_m[w][row] == -INFINITY ? 0.f : __expf(part_m[w][row] - mm);
        lt += part_l[w][row] * sc[w];
      }
      const float inv = lt > 0.f ? 1.0f / lt : 0.f;
      const int g = row / p.Nq;
      const int qb = first + g, t = row - g * p.Nq;
      f16x8 o8, l8, h8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = db * 8 + e;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += part_o[w][d][row] * sc[w];
        const float x = acc * inv;
        o8[e] = (f16)x;             // (plain rows: the compiler may fold the multiply into the conversion — one rounding)
        // [hi | lo | hi]: hi and the value lo is taken against MUST be the same conversion of the same f32 — with the fold
        // above hi was RN16(acc . inv) while lo was computed against RN16(RN32(acc . inv)): one element in ~10^4 sits between
        // the two, and its hi + lo was one f16 ulp off (round 5: tests/test_parity_mode_gpu.py, split form on fragment tiles)
        const f16 h16 = Elt<T>::from_f32(x);
        h8[e] = h16;
        l8[e] = Elt<T>::from_f32(x - (float)h16);
      }
      f16* const og = p.out + ((size_t)qb * p.Nq + t) * p.ldo + h * 64 + db * 8;
      if (p.out_mode >= 2) {       // [hi | lo | hi] planes (VIDIL_DT_SPLIT3; 3: hi | lo only)
        const int pl = p.ldo / 3;
        *(f16x8*)og = h8;
        *(f16x8*)(og + pl) = l8;
        if (p.out_mode == 2) *(f16x8*)(og + 2 * pl) = h8;
      } else {
        *(f16x8*)og = o8;
      }
    }
  }
}

// ------------------------------------------------------------------ direct kernel, ONE WAVE per (kv batch, head)
// rows <= 32, K / V in fragment tiles or in rows (any number of key tiles).  The four-wave kernel above splits a unit's key tiles over its
// waves, issues every load, computes, and merges the partials through LDS: nothing of a workgroup is in flight while it computes
// and merges, and the overlap has to come from the other workgroups of the CU — measured (round 5, 3,584 images x 12 heads x 197
// keys, developer ablation): the loads alone 355 us (6.1 TB/s), with the merge 371, with the arithmetic 455 (plain) / 551 (split
// Q and P).  Here a wave owns a whole unit: it walks the unit's key tiles with the online softmax, tile kt + D requested before
// tile kt is computed (a register ring of D + 1 fragment sets), so every resident wave has D tiles (8 KiB each) in flight at any
// time; no LDS, no barrier, no merge — the 32 rows leave straight from the accumulators.  QS: the split form (f32 queries and the
// probabilities as hi + lo, as above).  The summation order differs from the four-wave kernel's (one running (m, l, O) instead of
// four merged ones), so a shape takes ONE of the two at every batch size and in either K / V layout: every launch of <= 32 rows and
// more than one key tile comes here (a session's tiled and row-major forms give the same bits, as they did on the four-wave kernel).
template <typename T, bool QS, int D, bool TILED>
__global__ __launch_bounds__(256, (QS || !TILED) ? 2 : 3) void attn_direct1_kernel(const AttnP<T> p) {
  using f16 = T;
  using f16x8 = typename Elt<T>::x8;
  constexpr int NBUF = D + 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int unit = blockIdx.x * 4 + wave;            // (adjacent waves: adjacent heads of one kv batch — adjacent K / V blocks)
  if (unit >= p.H * p.n_kv) return;
  const int z = unit / p.H, h = unit - z * p.H;
  int bk, first, count;
  resolve_unit(p, z, bk, first, count);
  const int rows = count * p.Nq;
  if (rows <= 0) return;  // a kv batch nobody attends to
  const int nk = p.Nk;
  const RowInfo ri = row_info(p, l31, first, rows);
  const f16* kg = p.k + ((size_t)bk * p.H + h) * p.Tk_cap * 64;
  const f16* vg = p.vt + ((size_t)bk * p.H + h) * 64 * (size_t)(TILED ? p.Tk_cap : p.NP);
  const int ntiles = (nk + 31) >> 5;

  f16x8 kf[NBUF][4], vf[NBUF][2][2];
  auto load = [&](f16x8 (&kfb)[4], f16x8 (&vfb)[2][2], int kt) {
    // (rows / key groups past the last key are not fetched: they stay zero and are masked in the tile)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kfb[ks] = zero8<T>();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) vfb[dt][hb] = zero8<T>();
    if constexpr (TILED) {
      // fragment tiles: each instruction of the wave reads one contiguous KiB
      const f16* kt_base = kg + (size_t)kt * 2048 + (hi * 32 + l31) * 8;
      if (kt * 32 + l31 < nk) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kfb[ks] = *(const f16x8*)(kt_base + ks * 512);
      }
      const f16* vt_base = vg + (size_t)kt * 2048 + (hi * 32 + l31) * 8;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        if ((kt * 2 + hb) * 16 + 4 * hi < nk) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) vfb[dt][hb] = *(const f16x8*)(vt_base + (hb * 2 + dt) * 512);
        }
      }
    } else {
      // K rows [Tk_cap][64], V^T rows [64][NP] in vt_pos order: the same fragments, 32-byte pieces of 32 rows
      int krow = kt * 32 + l31;
      krow = krow < nk ? krow : nk - 1;  // clamped rows are masked in the tile
      const f16* kr = kg + (size_t)krow * 64 + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kfb[ks] = *(const f16x8*)(kr + ks * 16);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int blk0 = (kt * 2 + hb) * 16;
        if (blk0 < nk) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) vfb[dt][hb] = *(const f16x8*)(vg + (size_t)(dt * 32 + l31) * p.NP + blk0 + 8 * hi);
        }
      }
    }
  };
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < ntiles) load(kf[j], vf[j], j);

  f16x8 qf[4], ql[QS ? 4 : 1];
  if constexpr (QS) {
    const float* qg = p.q32 + ((size_t)ri.qb * p.Nq + ri.t) * p.ldq32 + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 a = *(const f32x4*)(qg + ks * 16) * p.q_scale, b = *(const f32x4*)(qg + ks * 16 + 4) * p.q_scale;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf[ks][e] = Elt<T>::from_f32(a[e]);
        ql[ks][e] = Elt<T>::from_f32(a[e] - (float)qf[ks][e]);
        qf[ks][4 + e] = Elt<T>::from_f32(b[e]);
        ql[ks][4 + e] = Elt<T>::from_f32(b[e] - (float)qf[ks][4 + e]);
      }
    }
  } else {
    const f16* qg = p.q + (((size_t)ri.qb * p.H + h) * p.Tq_cap + ri.t) * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qg + ks * 16);
  }
  int kmin = ri.klim;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int x = __shfl_xor(kmin, o, 64);
    kmin = x < kmin ? x : kmin;
  }
  float m = -INFINITY, l = 0.f;
  f32x16 O[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;

  auto compute = [&](const f16x8 (&kfb)[4], const f16x8 (&vfb)[2][2], int kt) {
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    if constexpr (QS) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) S = Elt<T>::mfma32(kfb[ks], ql[ks], S);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) S = Elt<T>::mfma32(kfb[ks], qf[ks], S);
    const bool tail = (kt * 32 + 32) > nk;  // tile reaches past the last key: V^T needs zeroing too
    const bool need_mask = (kt * 32 + 32) > kmin;
    auto vfrag = [&](int dt, int hb) {
      const int blk0 = (kt * 2 + hb) * 16;
      f16x8 v = vfb[dt][hb];
      if (tail) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (blk0 + 4 * hi + (e & 3) + 8 * (e >> 2) >= nk) v[e] = (f16)0.f;
      }
      return v;
    };
    if constexpr (QS) softmax_pv_tile_psplit<T>(S, kt * 32, ri.klim, need_mask, m, l, O, vfrag);
    else softmax_pv_tile<T, true, false>(S, kt * 32, ri.klim, need_mask, m, l, O, vfrag);
  };
#pragma unroll 1
  for (int kt0 = 0; kt0 < ntiles; kt0 += NBUF) {
#pragma unroll
    for (int j = 0; j < NBUF; ++j) {
      const int kt = kt0 + j;
      if (kt < ntiles) {                             // (wave-uniform)
        if (kt + D < ntiles) load(kf[(j + D) % NBUF], vf[(j + D) % NBUF], kt + D);
        compute(kf[j], vf[j], kt);
      }
    }
  }
  l += __shfl_xor(l, 32, 64);
  store_rows(p, ri, h, O, l > 0.f ? 1.0f / l : 0.f);
}

// ------------------------------------------------------------------ one wave, one key tile
// rows <= 32 and Nk <= 32 (decoder self-attention over the cached tokens): a single wave per (unit, head),
// operands straight from memory, no LDS and no barrier.
template <typename T>
__global__ __launch_bounds__(64) void attn_wave_kernel(const AttnP<T> p) {
  using f16 = T;
  using f16x8 = typename Elt<T>::x8;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  int bk, first, count;
  resolve_unit(p, blockIdx.z, bk, first, count);
  const int rows = count * p.Nq;
  if (rows <= 0) return;
  const int nk = p.Nk;
  const RowInfo ri = row_info(p, l31, first, rows);
  const f16* kg = p.k + ((size_t)bk * p.H + h) * p.Tk_cap * 64;
  const f16* vg = p.vt + ((size_t)bk * p.H + h) * 64 * (size_t)p.NP;
  const f16* qg = p.q + (((size_t)ri.qb * p.H + h) * p.Tq_cap + ri.t) * 64 + hi * 8;
  int krow = l31 < nk ? l31 : nk - 1;
  const f16* kr = kg + (size_t)krow * 64 + hi * 8;
  // every operand of the unit is requested before the first MFMA: K and Q fragments, and the four V^T fragments
  f16x8 kf[4], qf[4], vf[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *(const f16x8*)(kr + ks * 16);
    qf[ks] = *(const f16x8*)(qg + ks * 16);
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int blk0 = hb * 16;
      vf[dt][hb] = zero8<T>();
      if (blk0 < nk) vf[dt][hb] = *(const f16x8*)(vg + (size_t)(dt * 32 + l31) * p.NP + blk0 + 8 * hi);
    }
  f32x16 S;
#pragma unroll
  for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) S = Elt<T>::mfma32(kf[ks], qf[ks], S);
  float m = -INFINITY, l = 0.f;
  f32x16 O[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
  softmax_pv_tile<T>(S, 0, ri.klim, true, m, l, O, [&](int dt, int hb) {
    const int blk0 = hb * 16;
    f16x8 v = vf[dt][hb];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (blk0 + 4 * hi + (e & 3) + 8 * (e >> 2) >= nk) v[e] = (f16)0.f;      // keys past Nk: 0 * garbage must stay 0
    return v;
  });
  l += __shfl_xor(l, 32, 64);
  store_rows(p, ri, h, O, l > 0.f ? 1.0f / l : 0.f);
}

template <typename T, int NKT, int NW>
int launch_lds(const AttnP<T>& p, int max_rows, hipStream_t s) {
  constexpr int smem_kv = NKT * 32 * KROW * 2 + 64 * (NKT * 32 + 8) * 2;
  // + 2 KiB per wave for the output transposition (store_rows_lds) where a second workgroup still fits beside it
  constexpr bool ostage = 2 * (smem_kv + NW * 2048) <= 160 * 1024;
  constexpr int smem = smem_kv + (ostage ? NW * 2048 : 0);
  static std::atomic<unsigned long long> attr_set{0};   // (one bit per device that has the opt-in: vidil_lds_opt_in)
  auto kern = attn_lds_kernel<T, NKT, NW>;
  if (const int rc_ = vidil_lds_opt_in(attr_set, (const void*)kern, smem, "attention")) return rc_;
  // rows a little over one round of NW blocks (the ITM cross-attention: 8 captions x 35 tokens = 280 rows per
  // image with NW = 8): a second round in the same workgroup instead of a second workgroup that would stage the
  // unit's K/V again for a handful of rows
  AttnP<T> q = p;
  q.ostage = (ostage && p.ldo % 8 == 0 && ((uintptr_t)p.out & 15) == 0) ? 1 : 0;   // (16-byte stores)
  q.rb = (p.Nk <= NKT * 32 && max_rows > NW * 32 && max_rows <= NW * 32 + NW * 16) ? 2 : 1;
  dim3 grid((max_rows + NW * 32 * q.rb - 1) / (NW * 32 * q.rb), p.H, p.n_kv);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, s, q);
  VIDIL_CHECK_LAUNCH("attention");
  return VIDIL_OK;
}

// The streamed kernel serves the towers' self-attention: row-major V, one query batch per K/V batch, no length table or
// causal mask, 129..224 query rows and <= NKT*32 keys, 16-byte aligned operands, 16-bit or e4m3 output rows; everything else
// stays on the staged kernel.
template <typename T>
bool stream_eligible(const AttnP<T>& p, int max_rows) {
  const char* e = vidil_dev_env("VIDIL_ATTN_STREAM");
  if (e != nullptr && e[0] == '0') return false;
  return p.NP == 0 && p.kv_len == nullptr && p.kv_index == nullptr && p.group_start == nullptr && p.kv_group == 1 && !p.causal &&
         max_rows > 128 && max_rows <= 224 && p.Nk > 192 && !p.tiled && (p.out_mode == 0 ? p.ldo % 8 == 0 : (p.out_mode == 1 && p.ldo % 16 == 0)) &&
         (uint64_t)p.Nq * p.ldo * 2 < 0x80000000ull && (uint64_t)p.n_kv * p.H * p.Tq_cap * 128 < 0x100000000ull && (((uintptr_t)p.k | (uintptr_t)p.vt | (uintptr_t)p.q | (uintptr_t)p.out) & 15) == 0;
}

template <typename T, int NKT>
int launch_stream(const AttnP<T>& p, hipStream_t s) {
  constexpr int smem = NKT * (4096 + 4 * 1152) + 8 * 2048;   // (attn_stream_kernel: TILE)
  static_assert(2 * smem <= 160 * 1024, "two workgroups per CU");
  static std::atomic<unsigned long long> attr_set16{0}, attr_set8{0};   // (one bit per device that has the opt-in: vidil_lds_opt_in)
  const int n_cu = vidil_cu_count();
  auto kern16 = attn_stream_kernel<T, NKT, false>;
  auto kern8 = attn_stream_kernel<T, NKT, true>;
  if (const int rc_ = vidil_lds_opt_in(attr_set16, (const void*)kern16, smem, "attention (stream kernel)")) return rc_;
  if (const int rc_ = vidil_lds_opt_in(attr_set8, (const void*)kern8, smem, "attention (stream kernel)")) return rc_;
  AttnP<T> q = p;
  q.ostage = 1;
  // two resident workgroups per CU, every one walking the same number of units (+-1)
  const int units = p.n_kv * p.H, slots = 2 * n_cu;
  const int rounds = (units + slots - 1) / slots;
  const int grid = (units + rounds - 1) / rounds;
  if (p.out_mode == 1) hipLaunchKernelGGL(kern8, dim3(grid), dim3(512), smem, s, q);
  else hipLaunchKernelGGL(kern16, dim3(grid), dim3(512), smem, s, q);
  VIDIL_CHECK_LAUNCH("attention/stream");
  return VIDIL_OK;
}

// the one-wave-per-unit direct kernel; $VIDIL_ATTN_DIRECT1=0 keeps the four-wave kernel (developer A / B)
template <typename T>
bool direct1_enabled(const AttnP<T>& p) {
  const char* e = vidil_dev_env("VIDIL_ATTN_DIRECT1");
  return e == nullptr || e[0] != '0';
}
template <typename T, bool QS>
int launch_direct1(const AttnP<T>& p, hipStream_t s) {
  VIDIL_REQUIRE((long long)p.H * p.n_kv < 0x7fffffffLL, "attention: H=%d x %d kv batches overflow the unit index", p.H, p.n_kv);
  const int units = p.H * p.n_kv;
  const dim3 grid((units + 3) / 4);
  // prefetch depth (round 5, 3,584 images x 12 heads x 197 keys): plain 370 us at depth 1 (depth 2 spills: 588), split 402 at depth 1,
  // 413 at 2 ($VIDIL_ATTN_DIRECT1_DEPTH=2, split form only); the four-wave kernel 461 / 544
  if constexpr (QS) {
    const char* e = vidil_dev_env("VIDIL_ATTN_DIRECT1_DEPTH");
    if (e != nullptr && atoi(e) == 2 && p.tiled) {
      hipLaunchKernelGGL((attn_direct1_kernel<T, true, 2, true>), grid, dim3(256), 0, s, p);
      VIDIL_CHECK_LAUNCH("attention/direct1");
      return VIDIL_OK;
    }
  }
  if (p.tiled) hipLaunchKernelGGL((attn_direct1_kernel<T, QS, 1, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((attn_direct1_kernel<T, QS, 1, false>), grid, dim3(256), 0, s, p);
  VIDIL_CHECK_LAUNCH("attention/direct1");
  return VIDIL_OK;
}

template <typename T, int NKT>
int launch_any(const AttnP<T>& p, int max_rows, hipStream_t s) {
  if constexpr (NKT == 7)
    if (stream_eligible(p, max_rows)) return launch_stream<T, NKT>(p, s);
  if (max_rows <= 32 && NKT == 1 && !p.tiled) {
    hipLaunchKernelGGL(attn_wave_kernel<T>, dim3(1, p.H, p.n_kv), dim3(64), 0, s, p);
    VIDIL_CHECK_LAUNCH("attention/wave");
    return VIDIL_OK;
  }
  if (max_rows <= 32) {
    if (direct1_enabled(p)) return launch_direct1<T, false>(p, s);
    hipLaunchKernelGGL((attn_direct_kernel<T, NKT>), dim3(1, p.H, p.n_kv), dim3(256), 0, s, p);
    VIDIL_CHECK_LAUNCH("attention/direct");
    return VIDIL_OK;
  }
  if (max_rows > 128) return launch_lds<T, NKT, 8>(p, max_rows, s);
  return launch_lds<T, NKT, 4>(p, max_rows, s);
}

template <typename T>
int attention_dispatch(const AttnP<T>& p, int nkt, int max_rows, int Nk, hipStream_t s) {
  switch (nkt) {
    case 1: return launch_any<T, 1>(p, max_rows, s);
    case 2: return launch_any<T, 2>(p, max_rows, s);
    case 3: return launch_any<T, 3>(p, max_rows, s);
    case 4: return launch_any<T, 4>(p, max_rows, s);
    case 5: return launch_any<T, 5>(p, max_rows, s);
    case 6: return launch_any<T, 6>(p, max_rows, s);
    case 7: return launch_any<T, 7>(p, max_rows, s);
    case 8: return launch_any<T, 8>(p, max_rows, s);
    case 9: return launch_any<T, 9>(p, max_rows, s);
    default: break;
  }
  // longer sequences (577 tokens of a 384^2 ViT-B/16, 257+ of others): rounds of key tiles in the direct
  // kernel, 224-key chunks re-staged through LDS in the staged kernel
  if (nkt <= 24) {
    if (max_rows <= 32) {
      if (direct1_enabled(p)) return launch_direct1<T, false>(p, s);
      hipLaunchKernelGGL((attn_direct_kernel<T, 24>), dim3(1, p.H, p.n_kv), dim3(256), 0, s, p);
      VIDIL_CHECK_LAUNCH("attention/direct");
      return VIDIL_OK;
    }
    if (max_rows > 128) return launch_lds<T, 7, 8>(p, max_rows, s);
    return launch_lds<T, 7, 4>(p, max_rows, s);
  }
  vidil_set_error("attention: Nk=%d > 768 not supported by these kernels", Nk);
  return VIDIL_EUNSUP;
}

}  // namespace

extern "C" int vidil_attention(const void* q, const void* k, const void* vt, void* out, const int32_t* kv_len,
                               const int32_t* kv_index, const int32_t* group_start, int32_t n_kv, int32_t max_group,
                               int32_t Bq, int32_t H, int32_t Nq, int32_t Nk, int32_t Tq_cap, int32_t Tk_cap, int32_t NP,
                               int32_t kv_group, int32_t causal, int32_t causal_off, int32_t ldo, int32_t kv_tiled,
                               int32_t dtype, int32_t out_dtype, void* stream) {
  VIDIL_REQUIRE(q && k && vt && out, "attention: null pointer");
  VIDIL_REQUIRE(Bq > 0 && H > 0 && Nq > 0 && Nk > 0, "attention: bad shape Bq=%d H=%d Nq=%d Nk=%d", Bq, H, Nq, Nk);
  if (kv_tiled) {
    VIDIL_REQUIRE(Tq_cap >= Nq && Tk_cap >= Nk && Tk_cap % 32 == 0,
                  "attention: tiled K/V need Tk_cap=%d >= Nk=%d and a multiple of 32", Tk_cap, Nk);
    NP = Tk_cap;   // unused by the tiled loads; keeps the checks below meaningful
  }
  VIDIL_REQUIRE(Tq_cap >= Nq && Tk_cap >= Nk && (NP == 0 || NP >= Nk), "attention: capacities too small");
  VIDIL_REQUIRE(NP % 16 == 0, "attention: NP=%d must be a multiple of 16 (V^T rows hold whole 16-key blocks)", NP);
  VIDIL_REQUIRE(ldo >= H * 64 && ldo % 8 == 0, "attention: ldo=%d must be >= H*64 and a multiple of 8", ldo);
  VIDIL_REQUIRE(kv_group > 0, "attention: kv_group=%d", kv_group);
  int units, max_rows;
  if (group_start != nullptr) {
    VIDIL_REQUIRE(kv_index == nullptr, "attention: group_start and kv_index are mutually exclusive");
    VIDIL_REQUIRE(n_kv > 0 && max_group > 0, "attention: group_start needs n_kv and max_group");
    units = n_kv;
    max_rows = max_group * Nq;
  } else if (kv_index != nullptr) {
    units = Bq;
    max_rows = Nq;
  } else {
    VIDIL_REQUIRE(Bq % kv_group == 0, "attention: Bq=%d not a multiple of kv_group=%d", Bq, kv_group);
    units = Bq / kv_group;
    max_rows = kv_group * Nq;
  }
  VIDIL_REQUIRE(H <= 65535 && units <= 65535, "attention: grid too large (H=%d units=%d)", H, units);
  const int nkt = (Nk + 31) / 32;
  VIDIL_REQUIRE(!kv_tiled || max_rows <= 32, "attention: tiled K/V serve at most 32 query rows per unit (got %d)", max_rows);
  // NP == 0: `vt` holds V row-major [Bk][H][Tk_cap][64]; only the LDS-staged kernel transposes on the way in
  VIDIL_REQUIRE(NP != 0 || max_rows > 32, "attention: row-major V (NP == 0) needs more than 32 query rows per unit (got %d)",
                max_rows);
  const bool split3 = out_dtype == (dtype | VIDIL_DT_SPLIT3);
  VIDIL_REQUIRE(out_dtype == dtype || split3 || (out_dtype == VIDIL_DT_FP8 && max_rows > 32 && ldo % 16 == 0),
                "attention: out_dtype=%d must equal dtype=%d (optionally | VIDIL_DT_SPLIT3), or be fp8 with more than 32 query "
                "rows per unit (staged kernel)", out_dtype, dtype);
  VIDIL_REQUIRE(!split3 || (ldo % 24 == 0 && ldo / 3 >= H * 64), "attention: split3 output needs ldo=%d = 3 planes of >= H*64, "
                "each a multiple of 8", ldo);
  VIDIL_DISPATCH_DTYPE(dtype, "attention", {
    const AttnP<T> p{(const T*)q, (const T*)k, (const T*)vt, (T*)out, kv_len, kv_index, group_start, Bq, H, Nq, Nk,
                     Tq_cap, Tk_cap, NP, kv_group, causal, causal_off, ldo, units, out_dtype == VIDIL_DT_FP8 ? 1 : (split3 ? 2 : 0),
                     kv_tiled ? 1 : 0, 1};
    return attention_dispatch<T>(p, nkt, max_rows, Nk, (hipStream_t)stream);
  });
}

// ====================================================================== f32 attention (parity precision mode only)
// The error-compensated "parity" mode carries every GEMM operand to ~2^-21, and tests/probes/probe_attention_rounding.py
// shows that what is then left of the caption-logit error — 2.4e-4 of the logit scale — is exactly the 16-bit rounding of
// Q / K / V (and of the probabilities) inside the MFMA attention kernels above.  This is the attention of that mode since
// round 4: plain f32 VALU arithmetic on f32 Q / K / V read IN PLACE from the row-major outputs of the projection GEMMs
// (element (row, head h, d) at base + row * ld + off + h * 64 + d — no per-head scatter), so the mode's tolerance holds at
// a trained model's logit scale too.  It is a precision mode, not a throughput path: ~1/16 of the MFMA kernels' arithmetic
// rate (DESIGN.md §4 states the cost); softmax(q k^T * scale) v with the same grouping / masking forms as vidil_attention.
//
//   attn_f32_kernel        one workgroup (4 waves) per (unit, head, block of 32 virtual query rows): 64 keys x 64 dims of K and V
//                          at a time in LDS (row stride 68 floats: conflict-free ds_read_b128 across keys), each wave owns 8
//                          rows and works on 4 at once — scores with lane = key, online softmax (wave reductions), then
//                          P.V with lane = d through a wave-private P buffer.
//   attn_f32_arena_kernel  decode-step self-attention over an f32 KV arena through the ancestry table (one query row per
//                          beam row, <= Tcap keys): one wave per (row, head), lane = d, online softmax over the keys.
namespace {

struct AttnF32P {
  const float* q;
  const float* k;
  const float* v;
  void* out;
  long long ldq, ldk, ldv, ldo;
  int q_off, k_off, v_off;
  int out_mode;                // 0: f32 rows; 2: 16-bit [hi | lo | hi] rows in three planes ldo / 3 apart (VIDIL_DT_SPLIT3)
  int dtype16;
  const int32_t* kv_len;
  const int32_t* kv_index;
  const int32_t* group_start;
  int Bq, H, Nq, Nk, kv_rows, kv_group, causal, causal_off, n_kv;
  const int32_t* anc;          // arena form
  int anc_ld, arena_rows;
  float scale;
};

template <typename T16>
__device__ __forceinline__ void store_f32_row(const AttnF32P& p, size_t row, int h, int d, float v) {
  if (p.out_mode == 0) {
    ((float*)p.out)[row * p.ldo + h * 64 + d] = v;
  } else {
    const long long pl = p.ldo / 3;
    T16* o = (T16*)p.out + row * p.ldo + h * 64 + d;
    const T16 hi = Elt<T16>::from_f32(v);
    const T16 lo = Elt<T16>::from_f32(v - (float)hi);
    o[0] = hi;
    o[pl] = lo;
    if (p.out_mode != 3) o[2 * pl] = hi;
  }
}

constexpr int F32_KC = 64;      // keys per staged chunk
constexpr int F32_LD = 68;      // floats per staged K / V / Q row (64 + 4: ds_read_b128 of consecutive rows hit distinct bank groups)
constexpr int F32_RB = 32;      // virtual query rows per workgroup

template <typename T16>
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnF32P p) {
  __shared__ __attribute__((aligned(16))) float Ks[F32_KC * F32_LD];
  __shared__ __attribute__((aligned(16))) float Vs[F32_KC * F32_LD];
  __shared__ __attribute__((aligned(16))) float Qs[F32_RB * F32_LD];
  __shared__ __attribute__((aligned(16))) float Ps[4][4][F32_KC];     // [wave][row of the group of 4][key]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y;
  int bk, first, count;
  resolve_unit(p, blockIdx.z, bk, first, count);
  const int rows = count * p.Nq;
  const int base = blockIdx.x * F32_RB;
  if (base >= rows) return;
  // ---- the block's query rows (pre-scaled), once
  for (int i = tid; i < F32_RB * 16; i += 256) {
    const int r = i >> 4, c = i & 15;
    const RowInfo ri = row_info(p, base + r, first, rows);
    f32x4 qv = {0.f, 0.f, 0.f, 0.f};
    if (ri.valid) qv = *(const f32x4*)(p.q + ((size_t)ri.qb * p.Nq + ri.t) * p.ldq + p.q_off + h * 64 + c * 4);
    *(f32x4*)(Qs + r * F32_LD + c * 4) = qv * p.scale;
  }
  // this wave's 8 rows, in two groups of 4: running maximum / sum (lane-uniform) and the output accumulator (lane = d)
  float m[2][4], l[2][4], acc[2][4];
  int klim[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      m[g][r] = -INFINITY; l[g][r] = 0.f; acc[g][r] = 0.f;
      const RowInfo ri = row_info(p, base + wave * 8 + g * 4 + r, first, rows);
      klim[g][r] = ri.valid ? ri.klim : 0;
    }
  const float* kg = p.k + (size_t)bk * p.kv_rows * p.ldk + p.k_off + h * 64;
  const float* vg = p.v + (size_t)bk * p.kv_rows * p.ldv + p.v_off + h * 64;
  for (int k0 = 0; k0 < p.Nk; k0 += F32_KC) {
    __syncthreads();     // (the previous chunk is consumed; the first pass: Qs is written)
    for (int i = tid; i < F32_KC * 16; i += 256) {
      const int r = i >> 4, c = i & 15;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (k0 + r < p.Nk) {
        kv = *(const f32x4*)(kg + (size_t)(k0 + r) * p.ldk + c * 4);
        vv = *(const f32x4*)(vg + (size_t)(k0 + r) * p.ldv + c * 4);
      }
      *(f32x4*)(Ks + r * F32_LD + c * 4) = kv;
      *(f32x4*)(Vs + r * F32_LD + c * 4) = vv;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // (wave-uniform: a group whose rows are all past the unit's last row, or whose keys of this chunk are all masked, has
      //  nothing to add — the decode steps' cross-attention has 3 rows per unit: one group of one wave works, and the LDS
      //  pipe, which bounds this kernel, is left to it)
      const int kmax = max(max(klim[g][0], klim[g][1]), max(klim[g][2], klim[g][3]));
      if (kmax <= k0) continue;
      // ---- scores of 4 rows x 64 keys: lane = key
      float sc[4] = {0.f, 0.f, 0.f, 0.f};
      const float* krow = Ks + lane * F32_LD;
      const float* qrow = Qs + (wave * 8 + g * 4) * F32_LD;
#pragma unroll 4
      for (int c = 0; c < 16; ++c) {
        const f32x4 kv = *(const f32x4*)(krow + c * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 qv = *(const f32x4*)(qrow + r * F32_LD + c * 4);      // (same address in every lane: a broadcast)
          sc[r] = __builtin_fmaf(qv[0], kv[0], sc[r]);
          sc[r] = __builtin_fmaf(qv[1], kv[1], sc[r]);
          sc[r] = __builtin_fmaf(qv[2], kv[2], sc[r]);
          sc[r] = __builtin_fmaf(qv[3], kv[3], sc[r]);
        }
      }
      // ---- online softmax per row; probabilities of this chunk into the wave's P buffer
      float alpha[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = (k0 + lane < klim[g][r]) ? sc[r] : -INFINITY;
        const float mn = fmaxf(m[g][r], wave_max(s));
        const float msafe = mn == -INFINITY ? 0.f : mn;
        const float pr = expf(s - msafe);                 // (s = -inf -> 0)
        alpha[r] = expf(m[g][r] - msafe);                 // (m = -inf -> 0)
        l[g][r] = l[g][r] * alpha[r] + wave_sum(pr);
        m[g][r] = mn;
        Ps[wave][r][lane] = pr;
      }
      // ---- P.V of the chunk: lane = d
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[g][r] *= alpha[r];
      for (int j = 0; j < F32_KC; j += 4) {
        f32x4 pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = *(const f32x4*)(&Ps[wave][r][j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = Vs[(j + e) * F32_LD + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[g][r] = __builtin_fmaf(pv[r][e], v, acc[g][r]);
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const RowInfo ri = row_info(p, base + wave * 8 + g * 4 + r, first, rows);
      if (!ri.valid) continue;
      const float inv = l[g][r] > 0.f ? 1.0f / l[g][r] : 0.f;
      store_f32_row<T16>(p, (size_t)ri.qb * p.Nq + ri.t, h, lane, acc[g][r] * inv);
    }
}

// The same attention on the f32-input MATRIX instruction (v_mfma_f32_32x32x2_f32: exact f32 products and accumulation at the f32
// vector rate — but each operand value is read from LDS once per 32 x 32 tile instead of once per FMA, which is what bounded
// the VALU kernel above: ~10x on the towers' shapes).  One workgroup = 4 waves = 4 x 32 virtual query rows of a unit; a wave
// owns its 32 rows over ALL keys (no merge across waves) in the transposed forms of the 16-bit kernels: S^T = K.Q^T (lane =
// query row, 16 keys of the tile per lane: the online softmax is lane-local plus one cross-half shuffle) and O^T = V^T.P^T
// (lane = row, registers = d).  K and V tiles of 32 keys x 64 dims are staged in LDS for the 4 waves (registers -> LDS, the next
// tile's global loads in flight under the current tile's MFMAs); Q lives in registers (32 per lane).  The k-index of a K.Q^T
// step s pairs d = s (lanes 0-31) with d = s + 32 (lanes 32-63), so a lane's 32 operands are one contiguous half row
// (8 ds_read_b128); the k-index of a V^T.P^T step r pairs the keys that the two half-waves hold in accumulator register r,
// so P needs no permutation.
constexpr int F32M_LDK = 68, F32M_LDV = 72;     // floats per staged K / V row (bank-conflict-free b128 / b32 fragment reads)

template <typename T16>
__global__ __launch_bounds__(256) void attn_f32_mfma_kernel(const AttnF32P p) {
  __shared__ __attribute__((aligned(16))) float Ks[32 * F32M_LDK];
  __shared__ __attribute__((aligned(16))) float Vs[32 * F32M_LDV];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  int bk, first, count;
  resolve_unit(p, blockIdx.z, bk, first, count);
  const int rows = count * p.Nq;
  const int base = blockIdx.x * 128;
  if (base >= rows) return;
  const RowInfo ri = row_info(p, base + wave * 32 + l31, first, rows);       // this lane's query row
  const bool wave_live = base + wave * 32 < rows;                              // (uniform)
  // Q operand: q[s] = Q[row][hi * 32 + s] * scale
  float qf[32];
  {
    const float* qrow = p.q + ((size_t)ri.qb * p.Nq + ri.t) * p.ldq + p.q_off + h * 64 + hi * 32;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 v = *(const f32x4*)(qrow + c * 4);        // (row_info clamps invalid rows onto the last valid one)
      qf[c * 4 + 0] = v[0] * p.scale; qf[c * 4 + 1] = v[1] * p.scale; qf[c * 4 + 2] = v[2] * p.scale; qf[c * 4 + 3] = v[3] * p.scale;
    }
  }
  const float* kg = p.k + (size_t)bk * p.kv_rows * p.ldk + p.k_off + h * 64;
  const float* vg = p.v + (size_t)bk * p.kv_rows * p.ldv + p.v_off + h * 64;
  // staging: thread -> (key = tid / 8, two 16-byte chunks c = tid % 8 and c + 8) of the tile, for K and for V
  const int skey = tid >> 3, sc = tid & 7;
  f32x4 kr[2], vr[2];
  auto fetch = [&](int k0) {
    const int key = k0 + skey;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      kr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      vr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (key < p.Nk) {
        kr[j] = *(const f32x4*)(kg + (size_t)key * p.ldk + (sc + 8 * j) * 4);
        vr[j] = *(const f32x4*)(vg + (size_t)key * p.ldv + (sc + 8 * j) * 4);
      }
    }
  };
  float m = -INFINITY, l = 0.f;
  f32x16 O[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
  fetch(0);
  for (int k0 = 0; k0 < p.Nk; k0 += 32) {
    __syncthreads();                       // every wave is done with the previous tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *(f32x4*)(Ks + skey * F32M_LDK + (sc + 8 * j) * 4) = kr[j];
      *(f32x4*)(Vs + skey * F32M_LDV + (sc + 8 * j) * 4) = vr[j];
    }
    __syncthreads();
    if (k0 + 32 < p.Nk) fetch(k0 + 32);    // (in flight under this tile's MFMAs)
    if (!wave_live) continue;
    // ---- S^T[key][row] = sum_d K[key][d] Q[row][d]: 32 steps of k = 2 (d = s | s + 32)
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    const float* kp = Ks + l31 * F32M_LDK + hi * 32;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 kv = *(const f32x4*)(kp + c * 4);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[0], qf[c * 4 + 0], S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[1], qf[c * 4 + 1], S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[2], qf[c * 4 + 2], S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[3], qf[c * 4 + 3], S, 0, 0, 0);
    }
    // ---- online softmax: S[r] belongs to key k0 + (r & 3) + 8 * (r >> 2) + 4 * hi of this lane's row
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= ri.klim) S[r] = -INFINITY;
      mt = fmaxf(mt, S[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float msafe = mn == -INFINITY ? 0.f : mn;
    const float alpha = expf(m - msafe);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = expf(S[r] - msafe);
      ps += S[r];
    }
    ps += __shfl_xor(ps, 32, 64);
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
    // ---- O^T[d][row] += sum_key V[key][d] P[row][key]: step r pairs the keys the two half-waves hold in register r
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* vp = Vs + ((r & 3) + 8 * (r >> 2) + 4 * hi) * F32M_LDV + l31;
      O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], S[r], O[0], 0, 0, 0);
      O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], S[r], O[1], 0, 0, 0);
    }
  }
  if (!wave_live || !ri.valid) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  // O[dt][r]: d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi -> 4 consecutive d per register quad
  const size_t row = (size_t)ri.qb * p.Nq + ri.t;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int d = dt * 32 + rq * 8 + 4 * hi;
      const f32x4 v = {O[dt][rq * 4 + 0] * inv, O[dt][rq * 4 + 1] * inv, O[dt][rq * 4 + 2] * inv, O[dt][rq * 4 + 3] * inv};
      if (p.out_mode == 0) {
        *(f32x4*)((float*)p.out + row * p.ldo + h * 64 + d) = v;
      } else {
        using x4 = typename Elt<T16>::x4;
        const long long pl = p.ldo / 3;
        T16* o = (T16*)p.out + row * p.ldo + h * 64 + d;
        x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vh[e] = Elt<T16>::from_f32(v[e]);
          vl[e] = Elt<T16>::from_f32(v[e] - (float)vh[e]);
        }
        *(x4*)o = vh;
        *(x4*)(o + pl) = vl;
        if (p.out_mode != 3) *(x4*)(o + 2 * pl) = vh;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// SPLIT-OPERAND form (round 5, arith = 1): the same f32 Q / K / V read in place, but every operand of the two contractions is
// handed to the 16-bit matrix instruction as hi + lo (hi = T16(x), lo = T16(x - hi): 22 significant bits for f16) and each
// contraction is three MFMAs — a_hi.b_hi + a_lo.b_hi + a_hi.b_lo, f32 accumulation; the lo.lo term (2^-22 of the product) is
// dropped — instead of one 16x slower f32-input instruction: the trick the error-compensated GEMMs of the mode already use.
// tests/probes/probe_precision_design.py (CPU, the fp32 oracle with exactly this arithmetic injected): caption logits 1.0e-6 of the
// logit scale from the fp32 reference at trained-like statistics, where 16-bit operands give 1.7e-4 and the budget of "1e-3
// absolute" at max|logit| = 16 is 6.4e-5.  One workgroup = 4 waves x 32 virtual query rows of a unit, each wave over all keys
// (S^T = K.Q^T, lane-local online softmax, O^T = V^T.P^T — the 16-bit kernels' transposed forms); K / V tiles of 32 keys are
// split while they are staged (registers -> LDS as 16-bit hi and lo images, the next tile's global loads in flight under the
// MFMAs): K in the GEMMs' XOR-swizzled 128-byte rows (ds_read_b128 fragments), V row-major in [d / 16][32 keys][16 d] blocks
// that ds_read_b64_tr_b16 reads transposed (the streamed tower kernel's layout: no key permutation, no 2-byte scatter).
// 24 MFMAs of v_mfma_f32_32x32x16 per 32 x 32 tile where the f32-input form issues 64 instructions of twice the latency.
template <typename T16, int NW>      // NW waves = NW x 32 virtual query rows of a unit per workgroup (8: a tower's 197 rows in ONE workgroup,
#ifndef VIDIL_SPLIT_LB8
#define VIDIL_SPLIT_LB8 4
#endif
__global__ __launch_bounds__(NW * 64, NW == 8 ? VIDIL_SPLIT_LB8 : 3) void attn_split_kernel(const AttnF32P p) {   // its K / V staged once)
  using x8 = typename Elt<T16>::x8;
  using x4 = typename Elt<T16>::x4;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  constexpr int NT = NW * 64;
  constexpr int NCH = 1024 / NT;          // 16-byte f32 chunks of the K tile (and of the V tile) per thread: 4 or 2... (32 keys x 16 chunks)
  constexpr int VSUB = 1152;              // bytes between the four [32 keys][16 d] blocks of a V image (1 KiB + 128: the two blocks
                                          // a half-wave reads together sit 32 banks apart)
  constexpr int KIMG = 4096, VIMG = 4 * VSUB;
  __shared__ __attribute__((aligned(16))) char smem[2 * KIMG + 2 * VIMG];
  char* const Kh = smem;
  char* const Kl = smem + KIMG;
  char* const Vh = smem + 2 * KIMG;
  char* const Vl = Vh + VIMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  int bk, first, count;
  resolve_unit(p, blockIdx.z, bk, first, count);
  const int rows = count * p.Nq;
  const int base = blockIdx.x * (NW * 32);
  if (base >= rows) return;
  const RowInfo ri = row_info(p, base + wave * 32 + l31, first, rows);       // this lane's query row
  const bool wave_live = base + wave * 32 < rows;                              // (uniform)
  // hi + lo of four f32: hi = the value with its low 13 mantissa bits cleared (exactly a 16-bit float for f16 operands inside
  // f16's normal range; T16's own rounding otherwise), lo = T16(x - hi): x - hi is exact in f32, so hi + lo carries 21-22 bits
  auto split4 = [](const f32x4& a, x4& vh, x4& vl) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      vh[e] = Elt<T16>::from_f32(a[e]);
      vl[e] = Elt<T16>::from_f32(a[e] - (float)vh[e]);
    }
  };
  // Q operand of k-step ks: d = ks * 16 + hi * 8 + 0..7 of this lane's row, scaled, hi and lo
  x8 qh[4], ql[4];
  {
    const float* qrow = p.q + ((size_t)ri.qb * p.Nq + ri.t) * p.ldq + p.q_off + h * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 a = *(const f32x4*)(qrow + ks * 16) * p.scale;       // (row_info clamps invalid rows onto the last valid one)
      const f32x4 b = *(const f32x4*)(qrow + ks * 16 + 4) * p.scale;
      x4 ah, al, bh, bl;
      split4(a, ah, al);
      split4(b, bh, bl);
      qh[ks] = __builtin_shufflevector(ah, bh, 0, 1, 2, 3, 4, 5, 6, 7);
      ql[ks] = __builtin_shufflevector(al, bl, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
  const float* kg = p.k + (size_t)bk * p.kv_rows * p.ldk + p.k_off + h * 64;
  const float* vg = p.v + (size_t)bk * p.kv_rows * p.ldv + p.v_off + h * 64;
  // staging: chunk q = tid + j * NT of the tile's 512 16-byte f32 chunks (key = q / 16, chunk c = q % 16), for K and for V
  f32x4 kr[NCH / 2], vr[NCH / 2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < NCH / 2; ++j) {
      const int q = tid + j * NT;
      const int key = k0 + (q >> 4), c = q & 15;
      kr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      vr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (key < p.Nk) {
        kr[j] = *(const f32x4*)(kg + (size_t)key * p.ldk + c * 4);
        vr[j] = *(const f32x4*)(vg + (size_t)key * p.ldv + c * 4);
      }
    }
  };
  const int sw = (l31 >> 1) & 7;
  // transposed V reads: lane t of 16-lane group g supplies row (t >> 2), 8-byte chunk (t & 3) of its group's [4 keys][16 d] block
  const int vlane = ((lane >> 4) & 1) * VSUB + (hi * 4 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
  auto read_v = [&](const char* img, int dt, int hb) {
    const char* a = img + vlane + hb * 512 + dt * (2 * VSUB);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 256));
    const s16x8 both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(x8, both);
  };
  // smallest key limit of the wave's rows: tiles that end below it need no mask
  int kmin = ri.klim;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int x = __shfl_xor(kmin, o, 64);
    kmin = x < kmin ? x : kmin;
  }
  float m = -INFINITY, l = 0.f;
  f32x16 O[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
  fetch(0);
  for (int k0 = 0; k0 < p.Nk; k0 += 32) {
    __syncthreads();                       // every wave is done with the previous tile
#pragma unroll
    for (int j = 0; j < NCH / 2; ++j) {
      const int q = tid + j * NT;
      const int skey = q >> 4, c = q & 15;   // 4 d values: d = c * 4 ..
      x4 a, b;
      split4(kr[j], a, b);
      const int ko = skey * 128 + (((c >> 1) ^ ((skey >> 1) & 7)) << 4) + (c & 1) * 8;
      *(x4*)(Kh + ko) = a;
      *(x4*)(Kl + ko) = b;
      split4(vr[j], a, b);
      const int vo = (c >> 2) * VSUB + skey * 32 + (c & 3) * 8;
      *(x4*)(Vh + vo) = a;
      *(x4*)(Vl + vo) = b;
    }
    __syncthreads();
    if (k0 + 32 < p.Nk) fetch(k0 + 32);    // (in flight under this tile's MFMAs)
    if (!wave_live) continue;
    // ---- S^T[key][row]: corrections first, the leading product last
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = l31 * 128 + (((ks * 2 + hi) ^ sw) << 4);
      const x8 kh = *(const x8*)(Kh + off);
      const x8 kl = *(const x8*)(Kl + off);
      S = Elt<T16>::mfma32(kl, qh[ks], S);
      S = Elt<T16>::mfma32(kh, ql[ks], S);
      S = Elt<T16>::mfma32(kh, qh[ks], S);
    }
    // ---- online softmax: S[r] belongs to key k0 + (r & 3) + 8 * (r >> 2) + 4 * hi of this lane's row.  The reference m moves
    // only when a tile outgrows it by kLazy (a wave-uniform decision: softmax_tile above has the argument)
    if (k0 + 32 > kmin) {                  // (uniform: a tile that reaches past some row's limit)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= ri.klim) S[r] = -INFINITY;
      }
    }
    float mt = S[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, S[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const bool grow = mt > m + kLazy;      // (m == -inf: true as soon as the row has seen one finite score)
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
      const float mn = grow ? mt : m;
      const float alpha = __builtin_amdgcn_exp2f((m - (mn == -INFINITY ? 0.f : mn)) * kLog2e);   // keeps: 2^0; m == -inf: 0
      l *= alpha;
      m = mn;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
    }
    const float mc = (m == -INFINITY ? 0.f : m) * kLog2e;
    float ps = 0.f;
    x8 ph[2], pl[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], kLog2e, -mc));   // (masked: 2^-inf = 0)
      ps += e;
      const T16 eh = Elt<T16>::from_f32(e);
      ph[r >> 3][r & 7] = eh;
      pl[r >> 3][r & 7] = Elt<T16>::from_f32(e - (float)eh);
    }
    l += ps;
    // ---- O^T[d][row] += V^T[d][keys] . P^T[keys][row]
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const x8 vh = read_v(Vh, dt, hb);
        const x8 vl = read_v(Vl, dt, hb);
        O[dt] = Elt<T16>::mfma32(vl, ph[hb], O[dt]);
        O[dt] = Elt<T16>::mfma32(vh, pl[hb], O[dt]);
        O[dt] = Elt<T16>::mfma32(vh, ph[hb], O[dt]);
      }
  }
  if (!wave_live || !ri.valid) return;
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  // O[dt][r]: d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi -> 4 consecutive d per register quad
  const size_t row = (size_t)ri.qb * p.Nq + ri.t;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int d = dt * 32 + rq * 8 + 4 * hi;
      const f32x4 v = {O[dt][rq * 4 + 0] * inv, O[dt][rq * 4 + 1] * inv, O[dt][rq * 4 + 2] * inv, O[dt][rq * 4 + 3] * inv};
      if (p.out_mode == 0) {
        *(f32x4*)((float*)p.out + row * p.ldo + h * 64 + d) = v;
      } else {
        const long long pln = p.ldo / 3;
        T16* o = (T16*)p.out + row * p.ldo + h * 64 + d;
        x4 vh, vl;
        split4(v, vh, vl);
        *(x4*)o = vh;
        *(x4*)(o + pln) = vl;
        if (p.out_mode != 3) *(x4*)(o + 2 * pln) = vh;
      }
    }
}

template <typename T16>
__global__ __launch_bounds__(256) void attn_f32_arena_kernel(const AttnF32P p) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);       // (row, head)
  if (unit >= p.Bq * p.H) return;
  const int b = unit / p.H, h = unit - b * p.H;
  const float q = p.q[(size_t)b * p.ldq + p.q_off + h * 64 + lane] * p.scale;
  float m = -INFINITY, l = 0.f, acc = 0.f;
  for (int j = 0; j < p.Nk; ++j) {
    const size_t row = (size_t)j * p.arena_rows + p.anc[(size_t)b * p.anc_ld + j];
    const float s = wave_sum(q * p.k[row * p.ldk + p.k_off + h * 64 + lane]);
    const float mn = fmaxf(m, s);
    const float alpha = expf(m - mn), pr = expf(s - mn);
    l = l * alpha + pr;
    acc = __builtin_fmaf(pr, p.v[row * p.ldv + p.v_off + h * 64 + lane], acc * alpha);
    m = mn;
  }
  store_f32_row<T16>(p, (size_t)b, h, lane, acc / l);
}

// The arena form for up to 8 * MAXJ keys, in the shape of beam_attn_kernel (beam_attention.hip): one wave per (beam row, head);
// lane (g = lane >> 3, c = lane & 7) owns the 8 d of chunk c of keys g, g + 8, ... — every K / V load 16 bytes wide and all of
// them in flight together (their addresses depend on the ancestry row only), f32 scores / softmax / P.V on the VALU, the key
// groups combined by a reduce-scatter (7 shuffles).  The one-key-at-a-time kernel above it replaces took 250 us per decode-step
// launch at 10,752 beam rows (a wave reduction per key); the f32 arena holds twice the bytes of the 16-bit one, nothing more.
template <typename T16, int MAXJ>
__global__ __launch_bounds__(256) void attn_f32_arena_gather_kernel(const AttnF32P p) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);       // (row, head)
  if (unit >= p.Bq * p.H) return;
  const int b = unit / p.H, h = unit - b * p.H;
  const int g = lane >> 3, c = lane & 7;
  const int32_t* __restrict__ anc = p.anc + (size_t)b * p.anc_ld;
  const float* qp = p.q + (size_t)b * p.ldq + p.q_off + h * 64 + c * 8;
  const f32x4 q0 = *(const f32x4*)qp * p.scale, q1 = *(const f32x4*)(qp + 4) * p.scale;
  f32x4 k0[MAXJ], k1[MAXJ], v0[MAXJ], v1[MAXJ];
  bool ok[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int t = g + 8 * j;
    ok[j] = t < p.Nk;
    const int tc = ok[j] ? t : p.Nk - 1;          // lanes past the end re-read the last key (masked below)
    k0[j] = k1[j] = v0[j] = v1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (8 * j < p.Nk) {                            // (wave-uniform)
      const size_t row = (size_t)tc * p.arena_rows + anc[tc];
      const float* kp = p.k + row * p.ldk + p.k_off + h * 64 + c * 8;
      const float* vp = p.v + row * p.ldv + p.v_off + h * 64 + c * 8;
      k0[j] = *(const f32x4*)kp; k1[j] = *(const f32x4*)(kp + 4);
      v0[j] = *(const f32x4*)vp; v1[j] = *(const f32x4*)(vp + 4);
    }
  }
  float s[MAXJ], m = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) d = fmaf(q0[e], k0[j][e], d);
#pragma unroll
    for (int e = 0; e < 4; ++e) d = fmaf(q1[e], k1[j][e], d);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    s[j] = ok[j] ? d : -INFINITY;
    m = fmaxf(m, s[j]);
  }
  m = fmaxf(m, __shfl_xor(m, 8, 64));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));     // Nk >= 1: finite
  float o[8], l = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const float pj = ok[j] ? expf(s[j] - m) : 0.f;
    l += pj;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = fmaf(pj, v0[j][e], o[e]); o[4 + e] = fmaf(pj, v1[j][e], o[4 + e]); }
  }
  l += __shfl_xor(l, 8, 64);
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  // sum over the 8 key groups as a reduce-scatter; lane (g, c) ends with d = 8c + 4*g2 + 2*g1 + g0
  const bool b2 = (g & 4) != 0, b1 = (g & 2) != 0, b0 = (g & 1) != 0;
  float o4[4], o2[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float keep = b2 ? o[4 + i] : o[i], send = b2 ? o[i] : o[4 + i];
    o4[i] = keep + __shfl_xor(send, 32, 64);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float keep = b1 ? o4[2 + i] : o4[i], send = b1 ? o4[i] : o4[2 + i];
    o2[i] = keep + __shfl_xor(send, 16, 64);
  }
  const float keep = b0 ? o2[1] : o2[0], send = b0 ? o2[0] : o2[1];
  const float od = keep + __shfl_xor(send, 8, 64);
  const int d = 8 * c + (b2 ? 4 : 0) + (b1 ? 2 : 0) + (b0 ? 1 : 0);
  store_f32_row<T16>(p, (size_t)b, h, d, od / l);
}

}  // namespace

extern "C" int vidil_attention_f32(const vidil_attn_f32_args* a, void* stream) {
  VIDIL_REQUIRE(a && a->q && a->k && a->v && a->out, "attention_f32: null pointer");
  VIDIL_REQUIRE(a->Bq > 0 && a->H > 0 && a->Nq > 0 && a->Nk > 0, "attention_f32: bad shape Bq=%d H=%d Nq=%d Nk=%d", a->Bq, a->H, a->Nq, a->Nk);
  VIDIL_REQUIRE(a->ldq % 4 == 0 && a->ldk % 4 == 0 && a->ldv % 4 == 0 && a->q_off % 4 == 0 && a->k_off % 4 == 0 && a->v_off % 4 == 0 &&
                    ((uintptr_t)a->q & 15) == 0 && ((uintptr_t)a->k & 15) == 0 && ((uintptr_t)a->v & 15) == 0,
                "attention_f32: rows and head offsets must be 16-byte aligned");
  VIDIL_REQUIRE(a->out_mode == 0 || ((a->out_mode == 2 || a->out_mode == 3) && a->ldo % 3 == 0 && a->ldo / 3 >= (long long)a->H * 64 &&
                                     (a->dtype16 == VIDIL_DT_F16 || a->dtype16 == VIDIL_DT_BF16)),
                "attention_f32: out_mode 0 (f32 rows), 2 ([hi | lo | hi] 16-bit rows, ldo = 3 planes) or 3 (those planes, hi | lo written)");
  VIDIL_REQUIRE(a->out_mode != 0 || a->ldo >= (long long)a->H * 64, "attention_f32: ldo=%lld < H*64 (f32 rows)", (long long)a->ldo);
  AttnF32P p;
  p.q = a->q; p.k = a->k; p.v = a->v; p.out = a->out;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
  p.q_off = a->q_off; p.k_off = a->k_off; p.v_off = a->v_off;
  p.out_mode = a->out_mode; p.dtype16 = a->dtype16;
  p.kv_len = a->kv_len; p.kv_index = a->kv_index; p.group_start = a->group_start;
  p.Bq = a->Bq; p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk; p.kv_rows = a->kv_rows; p.kv_group = a->kv_group;
  p.causal = a->causal; p.causal_off = a->causal_off; p.n_kv = a->n_kv;
  p.anc = a->anc; p.anc_ld = a->anc_ld; p.arena_rows = a->arena_rows; p.scale = a->scale;
  hipStream_t s = (hipStream_t)stream;
  const bool bf = a->dtype16 == VIDIL_DT_BF16;
  if (a->anc != nullptr) {
    VIDIL_REQUIRE(a->Nq == 1 && a->arena_rows > 0 && a->anc_ld >= a->Nk, "attention_f32: the arena form serves one query row per batch");
    const int units = a->Bq * a->H;
    const char* eg = vidil_dev_env("VIDIL_ATTN_F32_ARENA_GATHER");       // (developer: 0 = the one-key-at-a-time kernel)
    const bool allow_gather = !(eg && eg[0] == '0');
    if (allow_gather && a->Nk <= 32 && a->ldq % 4 == 0) {     // (the decode steps of a caption search: max_length <= 32)
      if (bf) hipLaunchKernelGGL((attn_f32_arena_gather_kernel<bf16, 4>), dim3((units + 3) / 4), dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_f32_arena_gather_kernel<f16, 4>), dim3((units + 3) / 4), dim3(256), 0, s, p);
    } else if (bf) hipLaunchKernelGGL(attn_f32_arena_kernel<bf16>, dim3((units + 3) / 4), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_f32_arena_kernel<f16>, dim3((units + 3) / 4), dim3(256), 0, s, p);
    VIDIL_CHECK_LAUNCH("attention_f32 (arena)");
    return VIDIL_OK;
  }
  VIDIL_REQUIRE(a->kv_rows >= a->Nk && a->kv_group > 0, "attention_f32: kv_rows=%d < Nk=%d or kv_group=%d", a->kv_rows, a->Nk, a->kv_group);
  int units, max_rows;
  if (a->group_start != nullptr) {
    VIDIL_REQUIRE(a->kv_index == nullptr && a->n_kv > 0 && a->max_group > 0, "attention_f32: group_start needs n_kv and max_group (and no kv_index)");
    units = a->n_kv;
    max_rows = a->max_group * a->Nq;
  } else if (a->kv_index != nullptr) {
    units = a->Bq;
    max_rows = a->Nq;
  } else {
    VIDIL_REQUIRE(a->Bq % a->kv_group == 0, "attention_f32: Bq=%d not a multiple of kv_group=%d", a->Bq, a->kv_group);
    units = a->Bq / a->kv_group;
    max_rows = a->kv_group * a->Nq;
  }
  VIDIL_REQUIRE(a->H <= 65535 && units <= 65535, "attention_f32: grid too large (H=%d units=%d)", a->H, units);
  if (a->arith == 1 && a->kv16 != 0) {
    // split-operand form against 16-bit K / V fragment tiles (the decode steps' cross-attention): the direct kernel's QS form
    VIDIL_REQUIRE(max_rows <= 32 && a->kv_rows % 32 == 0 && a->kv_rows >= a->Nk && a->Nk <= 768,
                  "attention_f32 (kv16): at most 32 query rows per unit (got %d), kv_rows=%d a multiple of 32 and >= Nk=%d <= 768", max_rows,
                  a->kv_rows, a->Nk);
    VIDIL_REQUIRE(a->q_off == 0 && !a->causal && a->kv_len == nullptr, "attention_f32 (kv16): no q_off / causal / kv_len in this form");
    VIDIL_REQUIRE(a->out_mode >= 2 ? a->ldo % 24 == 0 : false, "attention_f32 (kv16): out_mode 2 / 3 ([hi | lo | hi] rows, planes a multiple of 8) only");
    const int nkt = (a->Nk + 31) / 32;
    VIDIL_DISPATCH_DTYPE(a->dtype16, "attention_f32 (kv16)", {
      AttnP<T> q{};
      q.k = (const T*)(const void*)a->k; q.vt = (const T*)(const void*)a->v; q.out = (T*)a->out;
      q.kv_index = a->kv_index; q.group_start = a->group_start;
      q.Bq = a->Bq; q.H = a->H; q.Nq = a->Nq; q.Nk = a->Nk; q.Tq_cap = a->Nq; q.Tk_cap = a->kv_rows; q.NP = a->kv_rows;
      q.kv_group = a->kv_group; q.ldo = (int)a->ldo; q.n_kv = units; q.out_mode = a->out_mode; q.tiled = 1; q.rb = 1;
      q.q32 = a->q; q.ldq32 = a->ldq; q.q_scale = a->scale;
      const dim3 g(1, a->H, units);
      if (direct1_enabled(q)) return launch_direct1<T, true>(q, s);
      switch (nkt) {
        case 1: hipLaunchKernelGGL((attn_direct_kernel<T, 1, true>), g, dim3(256), 0, s, q); break;
        case 2: hipLaunchKernelGGL((attn_direct_kernel<T, 2, true>), g, dim3(256), 0, s, q); break;
        case 3: hipLaunchKernelGGL((attn_direct_kernel<T, 3, true>), g, dim3(256), 0, s, q); break;
        case 4: hipLaunchKernelGGL((attn_direct_kernel<T, 4, true>), g, dim3(256), 0, s, q); break;
        case 5: hipLaunchKernelGGL((attn_direct_kernel<T, 5, true>), g, dim3(256), 0, s, q); break;
        case 6: hipLaunchKernelGGL((attn_direct_kernel<T, 6, true>), g, dim3(256), 0, s, q); break;
        case 7: {
          const char* ev = vidil_dev_env("VIDIL_ATTN_QS_VARIANT");
          const int var = ev ? atoi(ev) : 2;      // (one key tile per wave and round, three waves per SIMD: 638 -> 555 us per 3,584-image launch)
          // (variant 1 — two key tiles per wave at three waves per SIMD — spills 27 registers: 602 us; not built any more)
          if (var == 2) hipLaunchKernelGGL((attn_direct_kernel<T, 7, true, 2>), g, dim3(256), 0, s, q);
          else hipLaunchKernelGGL((attn_direct_kernel<T, 7, true>), g, dim3(256), 0, s, q);
          break;
        }
        case 8: hipLaunchKernelGGL((attn_direct_kernel<T, 8, true>), g, dim3(256), 0, s, q); break;
        default: hipLaunchKernelGGL((attn_direct_kernel<T, 24, true>), g, dim3(256), 0, s, q); break;
      }
      VIDIL_CHECK_LAUNCH("attention_f32 (kv16 direct)");
      return VIDIL_OK;
    });
  }
  VIDIL_REQUIRE(a->kv16 == 0, "attention_f32: kv16 needs arith == 1");
  if (a->arith == 1) {
    // split-operand form on f32 Q / K / V in place: any number of rows per unit (a unit of a few rows leaves three of the four
    // waves without rows: they still stage)
    VIDIL_REQUIRE((a->out_mode >= 2 ? a->ldo % 12 == 0 : a->ldo % 4 == 0) && ((uintptr_t)a->out & 15) == 0,
                  "attention_f32 (split): output rows must allow 8-byte (split3) / 16-byte (f32) stores");
    // 4 waves (128 rows) per workgroup. The 8-wave form (one staging of a unit's K / V serves up to 256 rows: $VIDIL_ATTN_SPLIT_NW=8)
    // measured 2 - 12 % slower on a tower's 197 rows at 512 ... 3584 images and 0.8 % slower end to end (round 5):
    // 59 of its 256 rows are padding, and the staging it saves was not what bounds the kernel
    const char* ew = vidil_dev_env("VIDIL_ATTN_SPLIT_NW");
    const int nw = ew ? atoi(ew) : 4;
    if (nw == 8) {
      const dim3 gridm((max_rows + 255) / 256, a->H, units);
      if (bf) hipLaunchKernelGGL((attn_split_kernel<bf16, 8>), gridm, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((attn_split_kernel<f16, 8>), gridm, dim3(512), 0, s, p);
    } else {
      const dim3 gridm((max_rows + 127) / 128, a->H, units);
      if (bf) hipLaunchKernelGGL((attn_split_kernel<bf16, 4>), gridm, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_split_kernel<f16, 4>), gridm, dim3(256), 0, s, p);
    }
    VIDIL_CHECK_LAUNCH("attention_f32 (split)");
    return VIDIL_OK;
  }
  VIDIL_REQUIRE(a->arith == 0, "attention_f32: arith=%d (0: f32, 1: split-operand)", a->arith);
  // units of more than 8 query rows (the towers, the ITM encoder, prompt passes): the f32-MFMA kernel, 128 rows per workgroup;
  // a few rows per unit (the decode steps' cross-attention: 3 beams per image): the VALU kernel, which skips idle row groups
  static const bool allow_mfma = [] { const char* e = getenv("VIDIL_ATTN_F32_MFMA"); return !(e && e[0] == '0'); }();
  if (allow_mfma && max_rows > 8 && (a->out_mode >= 2 ? a->ldo % 12 == 0 : a->ldo % 4 == 0) && ((uintptr_t)a->out & 15) == 0) {   // (16-B / 8-B row stores)
    const dim3 gridm((max_rows + 127) / 128, a->H, units);
    if (bf) hipLaunchKernelGGL(attn_f32_mfma_kernel<bf16>, gridm, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_f32_mfma_kernel<f16>, gridm, dim3(256), 0, s, p);
    VIDIL_CHECK_LAUNCH("attention_f32 (mfma)");
    return VIDIL_OK;
  }
  const dim3 grid((max_rows + F32_RB - 1) / F32_RB, a->H, units);
  if (bf) hipLaunchKernelGGL(attn_f32_kernel<bf16>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(attn_f32_kernel<f16>, grid, dim3(256), 0, s, p);
  VIDIL_CHECK_LAUNCH("attention_f32");
  return VIDIL_OK;
}
