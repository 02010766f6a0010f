// attention.hip — softmax(Q K^T) V for the short sequences on this path
// (ViT 197 keys, CLIP 50/77/257, MED text 1..35 queries over <=20 cached or 197
// image keys).  One workgroup = (128 query rows, one head, one query batch);
// each wave owns 32 query rows.
//
// gfx950 design:
//   * K ([keys][64], rows padded to 144 B) and V^T ([64][keys], row stride
//     keys+4 halfs) of the head are staged once in LDS; both strides are chosen
//     so the ds_read_b128 / ds_read_b64 fragment reads are bank-conflict free.
//   * scores are computed TRANSPOSED (S^T = K·Q^T, v_mfma_f32_32x32x16_f16) so
//     a lane holds every score of its own query row: the whole softmax is
//     in-register (one cross-half shuffle per reduction), no online rescale is
//     needed because all keys (<=288) fit in registers.
//   * the P registers feed the P·V MFMA A-operand directly; the key order
//     inside a 16-key MFMA step is the order the S^T layout leaves them in
//     ({0-3,8-11} / {4-7,12-15} per half-wave) and the V^T fragment reads use
//     the same order, so no permute is needed.
#include "common.h"

namespace {

struct AttnP {
  const f16* q;
  const f16* k;
  const f16* vt;
  f16* out;
  const int32_t* kv_len;
  const int32_t* kv_index;
  int Bq, H, Nq, Nk, Tq_cap, Tk_cap, NP, kv_group, causal, causal_off, ldo;
};

constexpr int KROW = 72;  // halfs per K row in LDS (64 + 8 pad)

template <int NKT>
__global__ __launch_bounds__(256) void attn_kernel(const AttnP p) {
  constexpr int NKEY = NKT * 32;
  constexpr int VROW = NKEY + 4;  // halfs per V^T row in LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* Ks = (f16*)smem;                       // [NKEY][KROW]
  f16* Vs = (f16*)(smem + NKEY * KROW * 2);   // [64][VROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, bq = blockIdx.z;
  const int bk = p.kv_index != nullptr ? p.kv_index[bq] : bq / p.kv_group;
  int kvlen = p.Nk;
  if (p.kv_len != nullptr) {
    const int v = p.kv_len[bq];
    kvlen = v < kvlen ? v : kvlen;
  }

  // ---- stage K -------------------------------------------------------------
  {
    const f16* kg = p.k + ((size_t)bk * p.H + h) * p.Tk_cap * 64;
#pragma unroll
    for (int it = 0; it < NKT; ++it) {
      const int q = it * 256 + tid;
      const int row = q >> 3, c = q & 7;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (row < kvlen) v = *(const f16x8*)(kg + (size_t)row * 64 + c * 8);
      *(f16x8*)(Ks + row * KROW + c * 8) = v;
    }
  }
  // ---- stage V^T -----------------------------------------------------------
  {
    const f16* vg = p.vt + ((size_t)bk * p.H + h) * 64 * (size_t)p.NP;
#pragma unroll
    for (int it = 0; it < NKT; ++it) {
      const int q = it * 256 + tid;
      const int d = q / (NKT * 4), kc = q - d * (NKT * 4);
      const int key0 = kc * 8;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (key0 < kvlen && key0 + 8 <= p.NP) {
        v = *(const f16x8*)(vg + (size_t)d * p.NP + key0);
        if (key0 + 8 > kvlen) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (key0 + e >= kvlen) v[e] = (f16)0.f;
        }
      } else if (key0 < kvlen) {  // ragged tail of the global row
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (key0 + e < kvlen && key0 + e < p.NP) v[e] = vg[(size_t)d * p.NP + key0 + e];
      }
      f16* dst = Vs + d * VROW + key0;
      *(f16x4*)(dst) = f16x4{v[0], v[1], v[2], v[3]};
      *(f16x4*)(dst + 4) = f16x4{v[4], v[5], v[6], v[7]};
    }
  }
  __syncthreads();

  const int q0 = blockIdx.x * 128 + wave * 32;
  if (q0 >= p.Nq) return;

  // ---- Q fragments (B operand: n = query = lane&31, k = d) ----------------------
  f16x8 qf[4];
  {
    int row = q0 + l31;
    row = row < p.Nq ? row : p.Nq - 1;
    const f16* qg = p.q + (((size_t)bq * p.H + h) * p.Tq_cap + row) * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qg + ks * 16);
  }

  // ---- S^T = K · Q^T ---------------------------------------------------------------
  f32x16 S[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f16x8 kf = *(const f16x8*)(Ks + (kt * 32 + l31) * KROW + ks * 16 + hi * 8);
      S[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], S[kt], 0, 0, 0);
    }
  }

  // ---- mask + softmax over this lane's query row ----------------------------------
  const int qpos = q0 + l31;
  int klim = kvlen;  // keys >= klim are excluded
  if (p.causal) {
    const int c = qpos + p.causal_off + 1;
    klim = c < klim ? c : klim;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float s = key < klim ? S[kt][r] : -INFINITY;
      S[kt][r] = s;
      mx = fmaxf(mx, s);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (mx == -INFINITY) mx = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(S[kt][r] - mx);
      S[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;

  // ---- O = P · V ---------------------------------------------------------------------
  f32x16 O[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
#pragma unroll
  for (int blk = 0; blk < 2 * NKT; ++blk) {
    const int kt = blk >> 1, hb = blk & 1;
    f16x8 pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[j] = (f16)S[kt][hb * 8 + j];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const f16* vrow = Vs + (dt * 32 + l31) * VROW + blk * 16 + 4 * hi;
      const f16x4 lo = *(const f16x4*)(vrow);
      const f16x4 up = *(const f16x4*)(vrow + 8);
      const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
      O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf, vf, O[dt], 0, 0, 0);
    }
  }

  // ---- store: O[dt][r] is (query = (r&3)+8*(r>>2)+4*hi, d = dt*32 + lane&31) -------
  f16* og = p.out + (size_t)bq * p.Nq * p.ldo + h * 64;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qq = (r & 3) + 8 * (r >> 2) + 4 * hi;
    const float il = __shfl(inv, qq, 64);
    const int row = q0 + qq;
    if (row < p.Nq) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) og[(size_t)row * p.ldo + dt * 32 + l31] = (f16)(O[dt][r] * il);
    }
  }
}

template <int NKT>
int launch(const AttnP& p, hipStream_t s) {
  constexpr int smem = NKT * 32 * KROW * 2 + 64 * (NKT * 32 + 4) * 2;
  static bool attr_set = false;
  auto kern = attn_kernel<NKT>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vidil_set_error("attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return VIDIL_ELAUNCH;
    }
    attr_set = true;
  }
  dim3 grid((p.Nq + 127) / 128, p.H, p.Bq);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
  VIDIL_CHECK_LAUNCH("attention");
  return VIDIL_OK;
}

}  // namespace

extern "C" int vidil_attention(const void* q, const void* k, const void* vt, void* out,
                               const int32_t* kv_len, const int32_t* kv_index, int32_t Bq, int32_t H, int32_t Nq,
                               int32_t Nk, int32_t Tq_cap, int32_t Tk_cap, int32_t NP,
                               int32_t kv_group, int32_t causal, int32_t causal_off,
                               int32_t ldo, void* stream) {
  VIDIL_REQUIRE(q && k && vt && out, "attention: null pointer");
  VIDIL_REQUIRE(Bq > 0 && H > 0 && Nq > 0 && Nk > 0, "attention: bad shape Bq=%d H=%d Nq=%d Nk=%d", Bq, H, Nq, Nk);
  VIDIL_REQUIRE(kv_group > 0 && (kv_index != nullptr || Bq % kv_group == 0), "attention: Bq=%d not a multiple of kv_group=%d", Bq, kv_group);
  VIDIL_REQUIRE(Tq_cap >= Nq && Tk_cap >= Nk && NP >= Nk, "attention: capacities too small");
  VIDIL_REQUIRE(NP % 8 == 0, "attention: NP=%d must be a multiple of 8", NP);
  VIDIL_REQUIRE(ldo >= H * 64, "attention: ldo=%d < H*64", ldo);
  VIDIL_REQUIRE(H <= 65535 && Bq <= 65535 * 1, "attention: grid too large (H=%d Bq=%d)", H, Bq);
  AttnP p{(const f16*)q, (const f16*)k, (const f16*)vt, (f16*)out, kv_len, kv_index, Bq, H, Nq, Nk,
          Tq_cap, Tk_cap, NP, kv_group, causal, causal_off, ldo};
  hipStream_t s = (hipStream_t)stream;
  const int nkt = (Nk + 31) / 32;
  switch (nkt) {
    case 1: return launch<1>(p, s);
    case 2: return launch<2>(p, s);
    case 3: return launch<3>(p, s);
    case 4: return launch<4>(p, s);
    case 5: return launch<5>(p, s);
    case 6: return launch<6>(p, s);
    case 7: return launch<7>(p, s);
    case 8: return launch<8>(p, s);
    case 9: return launch<9>(p, s);
    default: break;
  }
  vidil_set_error("attention: Nk=%d > 288 not supported by the in-register softmax kernel", Nk);
  return VIDIL_EUNSUP;
}
