// beam_attention.hip — cached self-attention of the caption decoder's decode steps, on an append-only KV
// arena with an ancestry table instead of a reordered cache.
//
// The reference keeps past_key_values per beam row and, every step, gathers the whole cache by beam_idx
// (models/med.py:951-955 _reorder_cache; HF generate() calls it once per step).  At 3072 beam rows x 12
// layers that is 2 x 1.8 GB of HBM traffic per step for data that does not change.  Here a key / value is
// written once, at [position][slot = the beam row that produced it][H*64] (EPI_ARENA of vidil_gemm_f16: a
// plain row-major store, no per-head scatter), and only the i32 table anc[row][position] -> slot is
// reordered (vidil_beam_ancestry, a few hundred KB).
//
// vidil_beam_attention: one wave per (beam row, head), one query token.  Lane (g = lane>>3, c = lane&7)
// owns d-chunk c (8 halfs = one 16-byte load) of keys g, g+8, g+16, ...: every K and V load is 16 B wide,
// all of them are in flight together (their addresses depend only on the ancestry row), the 8 lanes of a
// group cover one 128-byte K / V row, and no transposition is needed anywhere.  Scores, softmax and the
// P·V accumulation are f32 on the VALU (the f16 x f16 products are exact in f32): a decode step has one
// query row per (beam, head), so an MFMA tile would be 1/32 used and the kernel is bound by the gather.
#include "common.h"

namespace {

template <typename T>
struct BeamAttnP {
  const T* q;
  const T* k;
  const T* v;
  const int32_t* anc;
  T* out;
  int rows, H, n_keys, arena_rows, Tcap, ldo;
  int split3;   // out rows are [hi | lo | hi] planes ldo/3 apart (VIDIL_DT_SPLIT3)
};

template <typename T, int MAXJ>
__global__ __launch_bounds__(256) void beam_attn_kernel(const BeamAttnP<T> p) {
  using f16 = T;
  using f16x8 = typename Elt<T>::x8;
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);   // (row, head) unit of this wave
  if (w >= p.rows * p.H) return;
  const int r = w / p.H, h = w - r * p.H;
  const int g = lane >> 3, c = lane & 7;
  const size_t hd = (size_t)p.H * 64;
  const int32_t* __restrict__ anc = p.anc + (size_t)r * p.Tcap;

  const f16x8 qv = *(const f16x8*)(p.q + (size_t)r * hd + h * 64 + c * 8);
  size_t off[MAXJ];
  bool ok[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int t = g + 8 * j;
    ok[j] = t < p.n_keys;
    const int tc = ok[j] ? t : p.n_keys - 1;   // lanes past the end re-read the last key (masked below)
    off[j] = 0;
    if (8 * j < p.n_keys) off[j] = ((size_t)tc * p.arena_rows + anc[tc]) * hd + h * 64 + c * 8;   // wave-uniform guard
  }
  f16x8 kv[MAXJ], vv[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { kv[j][e] = (f16)0.f; vv[j][e] = (f16)0.f; }
    if (8 * j < p.n_keys) {
      kv[j] = *(const f16x8*)(p.k + off[j]);
      vv[j] = *(const f16x8*)(p.v + off[j]);
    }
  }

  // scores: the 8 lanes of a group each hold the partial dot product of their d-chunk
  float s[MAXJ];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d = fmaf((float)qv[e], (float)kv[j][e], d);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    s[j] = ok[j] ? d : -INFINITY;
    m = fmaxf(m, s[j]);
  }
  m = fmaxf(m, __shfl_xor(m, 8, 64));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));     // n_keys >= 1, so m is finite

  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const float pj = ok[j] ? __expf(s[j] - m) : 0.f;
    l += pj;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf(pj, (float)vv[j][e], o[e]);
  }
  l += __shfl_xor(l, 8, 64);
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);

  // sum over the 8 key groups as a reduce-scatter: every exchange halves the values a lane keeps
  // (7 shuffles instead of 24); lane (g, c) ends with d = 8c + 4*g2 + 2*g1 + g0.
  const bool b2 = (g & 4) != 0, b1 = (g & 2) != 0, b0 = (g & 1) != 0;
  float o4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float keep = b2 ? o[4 + i] : o[i];
    const float send = b2 ? o[i] : o[4 + i];
    o4[i] = keep + __shfl_xor(send, 32, 64);
  }
  float o2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float keep = b1 ? o4[2 + i] : o4[i];
    const float send = b1 ? o4[i] : o4[2 + i];
    o2[i] = keep + __shfl_xor(send, 16, 64);
  }
  const float keep = b0 ? o2[1] : o2[0];
  const float send = b0 ? o2[0] : o2[1];
  const float od = keep + __shfl_xor(send, 8, 64);
  const int d = 8 * c + (b2 ? 4 : 0) + (b1 ? 2 : 0) + (b0 ? 1 : 0);
  const float ov = od * (1.0f / l);
  const T oh = Elt<T>::from_f32(ov);
  T* const og = p.out + (size_t)r * p.ldo + h * 64 + d;
  og[0] = oh;
  if (p.split3) {
    const int pl = p.ldo / 3;
    og[pl] = Elt<T>::from_f32(ov - (float)oh);
    og[2 * pl] = oh;
  }
}

__global__ __launch_bounds__(256) void beam_ancestry_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                                            const int32_t* __restrict__ beam_idx, int rows, int Tcap,
                                                            int cur_pos) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * Tcap) return;
  const int r = i / Tcap, t = i - r * Tcap;
  int v = 0;
  if (t < cur_pos) {
    v = src[(size_t)beam_idx[r] * Tcap + t];
  } else if (t == cur_pos) {
    v = r;
  }
  dst[i] = v;
}

}  // namespace

extern "C" int vidil_beam_ancestry(const int32_t* anc_src, int32_t* anc_dst, const int32_t* beam_idx, int32_t rows,
                                   int32_t Tcap, int32_t cur_pos, void* stream) {
  VIDIL_REQUIRE(anc_src && anc_dst && beam_idx, "beam_ancestry: null pointer");
  VIDIL_REQUIRE(anc_src != anc_dst, "beam_ancestry: src and dst must be different buffers");
  VIDIL_REQUIRE(rows > 0 && Tcap > 0 && cur_pos >= 0 && cur_pos < Tcap, "beam_ancestry: bad shape rows=%d Tcap=%d cur_pos=%d",
                rows, Tcap, cur_pos);
  const long n = (long)rows * Tcap;
  VIDIL_REQUIRE(n < (1L << 31), "beam_ancestry: table too large");
  hipLaunchKernelGGL(beam_ancestry_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, anc_src,
                     anc_dst, beam_idx, rows, Tcap, cur_pos);
  VIDIL_CHECK_LAUNCH("beam_ancestry");
  return VIDIL_OK;
}

extern "C" int vidil_beam_attention(const void* q, const void* k_arena, const void* v_arena, const int32_t* anc, void* out,
                                    int32_t rows, int32_t H, int32_t n_keys, int32_t arena_rows, int32_t Tcap, int32_t ldo,
                                    int32_t dtype, int32_t out_dtype, void* stream) {
  VIDIL_REQUIRE(q && k_arena && v_arena && anc && out, "beam_attention: null pointer");
  VIDIL_REQUIRE(rows > 0 && H > 0 && n_keys > 0, "beam_attention: bad shape rows=%d H=%d n_keys=%d", rows, H, n_keys);
  VIDIL_REQUIRE(n_keys <= Tcap, "beam_attention: n_keys=%d exceeds the ancestry capacity Tcap=%d", n_keys, Tcap);
  VIDIL_REQUIRE(arena_rows >= rows, "beam_attention: arena_rows=%d < rows=%d", arena_rows, rows);
  VIDIL_REQUIRE(ldo >= H * 64, "beam_attention: ldo=%d must be >= H*64", ldo);
  const bool split3 = out_dtype == (dtype | VIDIL_DT_SPLIT3);
  VIDIL_REQUIRE(out_dtype == dtype || split3, "beam_attention: out_dtype=%d must be dtype=%d, optionally | VIDIL_DT_SPLIT3", out_dtype, dtype);
  VIDIL_REQUIRE(!split3 || (ldo % 3 == 0 && ldo / 3 >= H * 64), "beam_attention: split3 output needs ldo=%d = 3 planes of >= H*64", ldo);
  VIDIL_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k_arena & 15) == 0 && ((uintptr_t)v_arena & 15) == 0,
                "beam_attention: q / arenas must be 16-B aligned");
  VIDIL_REQUIRE(n_keys <= 64, "beam_attention: n_keys=%d > 64 not supported by this kernel", n_keys);
  const long units = (long)rows * H;
  const dim3 grid((unsigned)((units + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  VIDIL_DISPATCH_DTYPE(dtype, "beam_attention", {
    const BeamAttnP<T> p{(const T*)q, (const T*)k_arena, (const T*)v_arena, anc, (T*)out, rows, H, n_keys, arena_rows, Tcap, ldo, split3 ? 1 : 0};
    if (n_keys <= 32) {
      hipLaunchKernelGGL((beam_attn_kernel<T, 4>), grid, dim3(256), 0, s, p);
    } else {
      hipLaunchKernelGGL((beam_attn_kernel<T, 8>), grid, dim3(256), 0, s, p);
    }
  });
  VIDIL_CHECK_LAUNCH("beam_attention");
  return VIDIL_OK;
}
