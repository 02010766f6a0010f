// scan.hip — CLIP ontology scan: scores = img · txt^T per category with a fused
// per-frame top-k, so the [frames x 42,759] score matrix is never written to
// HBM or copied to the host (the reference does both:
// run_visual_tokenization.py:276,298-308).
//
// Exactness: top-k indices must be reproducible bit for bit, so the scan runs
// in f32 on the f32-input MFMA (v_mfma_f32_32x32x2_f32), which is an exact
// k-ordered fmaf chain.  A lane of half hi holds k = 8c + 4*hi + j (j = MFMA
// step 0..3 of chunk c), so every score is
//     s = 0;  for c: for j in 0..3:  s = fma(t[8c+j],   f[8c+j],   s);
//                                    s = fma(t[8c+4+j], f[8c+4+j], s);
// which oracle/scan_ref.c restates literally.
//
// Work split: grid = (class chunks, frame tiles of 32).  A workgroup keeps its
// 32 frame rows in LDS (padded, conflict-free b128 reads) and streams CT class
// tiles (32 classes each) of ONE category from HBM/L2 straight into MFMA A
// operands; each lane keeps a sorted top-k for (its frame, its classes) in
// registers; lists are merged through LDS and one partial list per
// (chunk, frame) goes to the workspace; a second tiny kernel merges chunks.
#include "common.h"

namespace {

constexpr int TOPK_MAX = 8;
constexpr int CT = 8;      // class tiles per workgroup
constexpr int MAXCAT = 8;

struct ScanP {
  const float* img;
  const float* txt;
  int NF, D, ncat, topk;
  int seg_start[MAXCAT], seg_len[MAXCAT];
  int chunk_prefix[MAXCAT + 1];  // chunks before category c
  float* part_s;   // [nchunks][NFpad][topk]
  int* part_i;
  int NFpad;
};

__device__ __forceinline__ bool better(float s1, int i1, float s2, int i2) {
  return s1 > s2 || (s1 == s2 && i1 < i2);
}

template <int TK>
__device__ __forceinline__ void insert(float (&ts)[TK], int (&ti)[TK], float s, int i) {
  if (better(s, i, ts[TK - 1], ti[TK - 1])) {
    ts[TK - 1] = s; ti[TK - 1] = i;
#pragma unroll
    for (int j = TK - 1; j > 0; --j) {
      if (better(ts[j], ti[j], ts[j - 1], ti[j - 1])) {
        const float a = ts[j]; ts[j] = ts[j - 1]; ts[j - 1] = a;
        const int b = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = b;
      }
    }
  }
}

template <int TK>
__global__ __launch_bounds__(256) void scan_kernel(const ScanP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int D = p.D;
  const int FROW = D + 4;  // floats per frame row in LDS
  float* Fs = (float*)smem;  // [32][FROW]; reused for the list merge afterwards

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int chunk = blockIdx.x, ftile = blockIdx.y;

  int cat = 0;
#pragma unroll
  for (int c = 1; c < MAXCAT; ++c)
    if (c < p.ncat && chunk >= p.chunk_prefix[c]) cat = c;
  const int tile0 = (chunk - p.chunk_prefix[cat]) * CT;            // first class tile within the category
  const int ntiles_cat = (p.seg_len[cat] + 31) / 32;
  const int seg_len = p.seg_len[cat];
  const float* txt = p.txt + (size_t)p.seg_start[cat] * D;

  // ---- stage the 32 frame rows ---------------------------------------------------
  {
    const int vec_per_row = D / 4;
    for (int q = tid; q < 32 * vec_per_row; q += 256) {
      const int r = q / vec_per_row, c = q - r * vec_per_row;
      const int f = ftile * 32 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (f < p.NF) v = *(const f32x4*)(p.img + (size_t)f * D + c * 4);
      *(f32x4*)(Fs + r * FROW + c * 4) = v;
    }
  }
  __syncthreads();

  float ts[TK];
  int ti[TK];
#pragma unroll
  for (int j = 0; j < TK; ++j) { ts[j] = -INFINITY; ti[j] = 0x7fffffff; }

  const float* frow = Fs + l31 * FROW + 4 * hi;
  for (int tt = wave; tt < CT; tt += 4) {
    const int tile = tile0 + tt;
    if (tile >= ntiles_cat) break;
    int cls = tile * 32 + l31;
    const int cls_ld = cls < seg_len ? cls : seg_len - 1;
    const float* trow = txt + (size_t)cls_ld * D + 4 * hi;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int c = 0; c < D / 8; ++c) {
      const f32x4 a = *(const f32x4*)(trow + c * 8);
      const f32x4 b = *(const f32x4*)(frow + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
    }
    // acc[r]: class = tile*32 + (r&3) + 8*(r>>2) + 4*hi ; frame = lane&31
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c2 = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (c2 < seg_len) insert<TK>(ts, ti, acc[r], c2);
    }
  }

  // ---- merge the 8 lists (4 waves x 2 halves) of each frame ------------------------
  __syncthreads();  // everyone is done reading Fs
  float* ms = (float*)smem;                       // [8][32][TK]
  int* mi = (int*)(smem + 8 * 32 * TK * 4);
  {
    const int slot = wave * 2 + hi;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      ms[(slot * 32 + l31) * TK + j] = ts[j];
      mi[(slot * 32 + l31) * TK + j] = ti[j];
    }
  }
  __syncthreads();
  if (tid < 32) {
    float bs[TK];
    int bi[TK];
#pragma unroll
    for (int j = 0; j < TK; ++j) { bs[j] = -INFINITY; bi[j] = 0x7fffffff; }
    for (int slot = 0; slot < 8; ++slot)
#pragma unroll
      for (int j = 0; j < TK; ++j) insert<TK>(bs, bi, ms[(slot * 32 + tid) * TK + j], mi[(slot * 32 + tid) * TK + j]);
    const size_t o = ((size_t)chunk * p.NFpad + ftile * 32 + tid) * p.topk;
#pragma unroll
    for (int j = 0; j < TK; ++j)
      if (j < p.topk) { p.part_s[o + j] = bs[j]; p.part_i[o + j] = bi[j]; }
  }
}

template <int TK>
__global__ void scan_merge_kernel(const ScanP p, int32_t* __restrict__ out_i, float* __restrict__ out_s) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.NF * p.ncat) return;
  const int f = g / p.ncat, cat = g - f * p.ncat;
  float bs[TK];
  int bi[TK];
#pragma unroll
  for (int j = 0; j < TK; ++j) { bs[j] = -INFINITY; bi[j] = 0x7fffffff; }
  for (int chunk = p.chunk_prefix[cat]; chunk < p.chunk_prefix[cat + 1]; ++chunk) {
    const size_t o = ((size_t)chunk * p.NFpad + f) * p.topk;
    for (int j = 0; j < p.topk; ++j) insert<TK>(bs, bi, p.part_s[o + j], p.part_i[o + j]);
  }
  for (int j = 0; j < p.topk; ++j) {
    out_i[((size_t)f * p.ncat + cat) * p.topk + j] = bi[j] == 0x7fffffff ? -1 : bi[j];
    out_s[((size_t)f * p.ncat + cat) * p.topk + j] = bs[j];
  }
}

// Dense variant for the BLIP retrieval backend (--encoder_version blip needs the k_test = 128 best texts per frame,
// too many for per-lane register lists): the same exact-f32 score of every (frame, text row), written out.
__global__ __launch_bounds__(256) void scores_kernel(const float* __restrict__ img, const float* __restrict__ txt, int NF, int D,
                                                     int NC, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int FROW = D + 4;
  float* Fs = (float*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int ftile = blockIdx.y;
  {
    const int vec_per_row = D / 4;
    for (int q = tid; q < 32 * vec_per_row; q += 256) {
      const int r = q / vec_per_row, c = q - r * vec_per_row;
      const int f = ftile * 32 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (f < NF) v = *(const f32x4*)(img + (size_t)f * D + c * 4);
      *(f32x4*)(Fs + r * FROW + c * 4) = v;
    }
  }
  __syncthreads();
  const float* frow = Fs + l31 * FROW + 4 * hi;
  const int f = ftile * 32 + l31;
  const int ntiles = (NC + 31) / 32;
  for (int tt = wave; tt < CT; tt += 4) {
    const int tile = blockIdx.x * CT + tt;
    if (tile >= ntiles) break;
    const int cls = tile * 32 + l31;
    const float* trow = txt + (size_t)(cls < NC ? cls : NC - 1) * D + 4 * hi;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int c = 0; c < D / 8; ++c) {
      const f32x4 a = *(const f32x4*)(trow + c * 8);
      const f32x4 b = *(const f32x4*)(frow + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
    }
    if (f < NF) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c2 = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (c2 < NC) out[(size_t)f * NC + c2] = acc[r];
      }
    }
  }
}

int total_chunks(int ncat, const int32_t* seg_len, int* prefix) {
  int n = 0;
  for (int c = 0; c < ncat; ++c) {
    if (prefix) prefix[c] = n;
    const int tiles = (seg_len[c] + 31) / 32;
    n += (tiles + CT - 1) / CT;
  }
  if (prefix) prefix[ncat] = n;
  return n;
}

}  // namespace

extern "C" int64_t vidil_scan_topk_ws_bytes(int32_t NF, int32_t NCpad, int32_t topk) {
  if (NF <= 0 || NCpad <= 0 || topk <= 0) return 0;
  // upper bound on chunks: one per CT*32 classes plus one ragged chunk per category
  const int64_t chunks = (int64_t)NCpad / (CT * 32) + MAXCAT + 1;
  const int64_t nfpad = ((int64_t)NF + 31) / 32 * 32;
  return chunks * nfpad * topk * 8;
}

extern "C" int vidil_scan_topk(const float* img, const float* txt, int32_t NF, int32_t D, int32_t ncat,
                               const int32_t* seg_start_host, const int32_t* seg_len_host, int32_t topk, void* partial,
                               int32_t* out_index, float* out_score, void* stream) {
  VIDIL_REQUIRE(img && txt && seg_start_host && seg_len_host && partial && out_index && out_score, "scan_topk: null pointer");
  VIDIL_REQUIRE(NF > 0 && D >= 8 && D % 8 == 0 && D <= 1024, "scan_topk: NF=%d D=%d (D%%8==0, D<=1024)", NF, D);
  VIDIL_REQUIRE(ncat >= 1 && ncat <= MAXCAT, "scan_topk: ncat=%d (1..%d)", ncat, MAXCAT);
  VIDIL_REQUIRE(topk >= 1 && topk <= TOPK_MAX, "scan_topk: topk=%d (1..%d)", topk, TOPK_MAX);
  ScanP p;
  p.img = img; p.txt = txt; p.NF = NF; p.D = D; p.ncat = ncat; p.topk = topk;
  for (int c = 0; c < MAXCAT; ++c) { p.seg_start[c] = 0; p.seg_len[c] = 0; }
  for (int c = 0; c < ncat; ++c) {
    VIDIL_REQUIRE(seg_len_host[c] > 0 && seg_start_host[c] >= 0 && seg_start_host[c] % 32 == 0,
                  "scan_topk: category %d: start=%d (must be a multiple of 32) len=%d", c, seg_start_host[c], seg_len_host[c]);
    p.seg_start[c] = seg_start_host[c];
    p.seg_len[c] = seg_len_host[c];
  }
  const int nchunks = total_chunks(ncat, seg_len_host, p.chunk_prefix);
  for (int c = ncat + 1; c <= MAXCAT; ++c) p.chunk_prefix[c] = nchunks;
  p.NFpad = (NF + 31) / 32 * 32;
  p.part_s = (float*)partial;
  p.part_i = (int*)((char*)partial + (size_t)nchunks * p.NFpad * topk * 4);
  hipStream_t s = (hipStream_t)stream;
  const int ftiles = p.NFpad / 32;
  VIDIL_REQUIRE(ftiles <= 65535, "scan_topk: too many frames in one call (%d)", NF);
  const int merge_bytes = 8 * 32 * TOPK_MAX * 8;
  int smem = 32 * (D + 4) * 4;
  if (smem < merge_bytes) smem = merge_bytes;
  auto kern = topk <= 5 ? scan_kernel<5> : scan_kernel<TOPK_MAX>;
  auto mkern = topk <= 5 ? scan_merge_kernel<5> : scan_merge_kernel<TOPK_MAX>;
  static int attr_smem = 0;
  if (smem > attr_smem) {
    hipError_t e1 = hipFuncSetAttribute((const void*)scan_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipError_t e2 = hipFuncSetAttribute((const void*)scan_kernel<TOPK_MAX>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      vidil_set_error("scan_topk: hipFuncSetAttribute failed");
      return VIDIL_ELAUNCH;
    }
    attr_smem = smem;
  }
  hipLaunchKernelGGL(kern, dim3(nchunks, ftiles), dim3(256), smem, s, p);
  VIDIL_CHECK_LAUNCH("scan_topk");
  const int nthreads = NF * ncat;
  hipLaunchKernelGGL(mkern, dim3((nthreads + 127) / 128), dim3(128), 0, s, p, out_index, out_score);
  VIDIL_CHECK_LAUNCH("scan_topk/merge");
  return VIDIL_OK;
}

extern "C" int vidil_scan_scores(const float* img, const float* txt, int32_t NF, int32_t D, int32_t NC, float* out,
                                 void* stream) {
  VIDIL_REQUIRE(img && txt && out, "scan_scores: null pointer");
  VIDIL_REQUIRE(NF > 0 && NC > 0 && D >= 8 && D % 8 == 0 && D <= 1024, "scan_scores: NF=%d NC=%d D=%d (D%%8==0, D<=1024)", NF, NC, D);
  const int ftiles = (NF + 31) / 32;
  VIDIL_REQUIRE(ftiles <= 65535, "scan_scores: too many frames in one call (%d)", NF);
  const int smem = 32 * (D + 4) * 4;
  static int attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute((const void*)scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
      vidil_set_error("scan_scores: hipFuncSetAttribute failed");
      return VIDIL_ELAUNCH;
    }
    attr_smem = smem;
  }
  const int chunks = ((NC + 31) / 32 + CT - 1) / CT;
  hipLaunchKernelGGL(scores_kernel, dim3(chunks, ftiles), dim3(256), smem, (hipStream_t)stream, img, txt, NF, D, NC, out);
  VIDIL_CHECK_LAUNCH("scan_scores");
  return VIDIL_OK;
}
