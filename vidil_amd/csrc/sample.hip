// sample.hip — one step of nucleus sampling for the caption decoder, on the device.
//
// Replaces: HF transformers 4.15 `sample()` as configured by models/blip.py:140-151 and
// run_video_CapFilt.py:103-104 — generate(do_sample=True, top_p=0.9, repetition_penalty=1.1, min_length,
// max_length, num_return_sequences=1) on a BertConfig whose default top_k is 50.  Per row and step:
//   1. RepetitionPenaltyLogitsProcessor: every distinct token already in the sequence (prompt included):
//      s = s < 0 ? s * p : s / p;
//   2. MinLengthLogitsProcessor: cur_len < min_length  =>  s[eos] = -inf;
//   3. TopKLogitsWarper(top_k): drop everything below the k-th largest score (ties with it are kept);
//   4. TopPLogitsWarper(top_p): candidates in descending order, drop candidate i (i >= 1) when the
//      cumulative probability of candidates 0..i-1 already exceeds top_p;
//   5. multinomial draw from the softmax of what is left; finished rows emit pad; a drawn eos finishes the row.
// torch.multinomial's CUDA generator stream cannot be reproduced, so the draw uses this library's own
// counter-based generator (Philox4x32-10, key = seed, counter = (row, step)) and inverse-CDF sampling over the
// candidates ordered by (score descending, token id ascending): the result is a deterministic function of
// (logits, sequence, seed, row, step), restated bit for bit by oracle/sample_ref.py.
//
// One workgroup per row: the row of V f32 logits is copied into LDS (122 KB for BERT's 30,524-token vocabulary),
// the top candidates are extracted by repeated block-wide arg-max (k is 50), thread 0 finishes the draw.
#include "common.h"

namespace {

constexpr int MAX_CAND = 64;

struct SampleP {
  const float* logits;   // [B][V]
  int32_t* seqs;         // [B][max_len]
  int32_t* done;         // [B]
  int32_t* n_done;       // [1]
  int32_t* next_tok;     // [B]
  int B, V, max_len, cur_len, min_length, eos, pad, top_k;
  float top_p, rep_penalty;
  uint32_t seed_lo, seed_hi, step;
  uint32_t row_offset;
};

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// uniform in [0, 1) with 24 random bits
__device__ __forceinline__ float philox_uniform(uint32_t seed_lo, uint32_t seed_hi, uint32_t row, uint32_t step) {
  uint32_t c[4] = {row, step, 0u, 0u};
  uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return (float)(c[0] >> 8) * (1.0f / 16777216.0f);
}

struct Best {
  float v;
  int i;
};
__device__ __forceinline__ Best better(Best a, Best b) {   // larger value, then lower index
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

__global__ __launch_bounds__(256) void sample_kernel(const SampleP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* row = (float*)smem;                 // [V]
  __shared__ Best wbest[4];
  __shared__ float cand_v[MAX_CAND];
  __shared__ int cand_i[MAX_CAND];
  __shared__ int n_cand, stop;
  const int b = blockIdx.x, tid = threadIdx.x;
  int32_t* seq = p.seqs + (size_t)b * p.max_len;
  if (p.done[b]) {                            // uniform per block
    if (tid == 0) {
      p.next_tok[b] = p.pad;
      seq[p.cur_len] = p.pad;
    }
    return;
  }
  const float* lg = p.logits + (size_t)b * p.V;
  for (int i = tid; i < p.V; i += 256) row[i] = lg[i];
  if (tid == 0) { n_cand = 0; stop = 0; }
  __syncthreads();
  // 1. repetition penalty (from the ORIGINAL value, so a token that occurs twice is penalised once)
  if (tid < p.cur_len) {
    const int t = seq[tid];
    if (t >= 0 && t < p.V) {
      const float s = lg[t];
      row[t] = s < 0.f ? s * p.rep_penalty : s / p.rep_penalty;
    }
  }
  __syncthreads();
  // 2. min length
  if (tid == 0 && p.cur_len < p.min_length && p.eos >= 0 && p.eos < p.V) row[p.eos] = -INFINITY;
  __syncthreads();
  // 3. top-k by repeated arg-max; ties with the k-th value are kept (up to MAX_CAND candidates)
  for (int it = 0; it < MAX_CAND; ++it) {
    Best me{-INFINITY, 0x7fffffff};
    for (int i = tid; i < p.V; i += 256) me = better(me, Best{row[i], i});
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      Best other{__shfl_xor(me.v, o, 64), __shfl_xor(me.i, o, 64)};
      me = better(me, other);
    }
    if ((tid & 63) == 0) wbest[tid >> 6] = me;
    __syncthreads();
    if (tid == 0) {
      Best w = better(better(wbest[0], wbest[1]), better(wbest[2], wbest[3]));
      const bool in_k = it < p.top_k;
      if (w.v == -INFINITY || (!in_k && w.v < cand_v[p.top_k - 1])) {
        stop = 1;
      } else {
        cand_v[it] = w.v;
        cand_i[it] = w.i;
        n_cand = it + 1;
        row[w.i] = -INFINITY;
      }
    }
    __syncthreads();
    if (stop) break;
  }
  // 4 + 5. nucleus cut and the draw (a few dozen candidates: one thread)
  if (tid == 0) {
    const int n = n_cand;
    int tok = p.eos;                           // unreachable fallback: n >= 1 whenever any logit is finite
    if (n > 0) {
      float c[MAX_CAND];
      float run = 0.f;
      const float v0 = cand_v[0];
      for (int i = 0; i < n; ++i) {
        run = __fadd_rn(run, __expf(cand_v[i] - v0));
        c[i] = run;
      }
      const float cut = __fmul_rn(p.top_p, run);
      int kept = n;
      for (int i = 0; i < n; ++i)
        if (c[i] > cut) { kept = i + 1; break; }   // candidate i is the first whose cumulative mass exceeds top_p: keep it, drop the rest
      const float u = philox_uniform(p.seed_lo, p.seed_hi, p.row_offset + (uint32_t)b, p.step);
      const float r = __fmul_rn(u, c[kept - 1]);
      int pick = kept - 1;
      for (int i = 0; i < kept; ++i)
        if (c[i] > r) { pick = i; break; }
      tok = cand_i[pick];
    }
    p.next_tok[b] = tok;
    seq[p.cur_len] = tok;
    if (tok == p.eos) {
      p.done[b] = 1;
      atomicAdd(p.n_done, 1);
    }
  }
}

}  // namespace

extern "C" int vidil_sample_top_k_top_p(const float* logits, int32_t* seqs, int32_t* done, int32_t* n_done, int32_t* next_tok,
                                        int32_t B, int32_t V, int32_t max_len, int32_t cur_len, int32_t min_length,
                                        int32_t eos_id, int32_t pad_id, int32_t top_k, float top_p, float rep_penalty,
                                        uint64_t seed, int32_t step, int32_t row_offset, void* stream) {
  VIDIL_REQUIRE(logits && seqs && done && n_done && next_tok, "sample: null pointer");
  VIDIL_REQUIRE(B > 0 && V > 0 && max_len > 0 && cur_len > 0 && cur_len < max_len, "sample: bad shape B=%d V=%d cur_len=%d max_len=%d",
                B, V, cur_len, max_len);
  VIDIL_REQUIRE(cur_len <= 256, "sample: cur_len=%d > 256 (one thread per previous token)", cur_len);
  VIDIL_REQUIRE(top_k >= 1 && top_k <= MAX_CAND - 14, "sample: top_k=%d outside 1..%d", top_k, MAX_CAND - 14);
  VIDIL_REQUIRE(top_p > 0.f && top_p <= 1.f && rep_penalty > 0.f, "sample: top_p=%g rep_penalty=%g", (double)top_p, (double)rep_penalty);
  const size_t lds = (size_t)V * 4;
  if (lds > 150 * 1024) {
    vidil_set_error("sample: vocabulary of %d tokens does not fit the LDS row buffer (<= 38400)", V);
    return VIDIL_EUNSUP;
  }
  static std::atomic<unsigned long long> attr_set{0};   // (one bit per device that has the opt-in: vidil_lds_opt_in)
  if (const int rc_ = vidil_lds_opt_in(attr_set, (const void*)sample_kernel, 150 * 1024, "sample")) return rc_;
  const SampleP p{logits, seqs, done, n_done, next_tok, B, V, max_len, cur_len, min_length, eos_id, pad_id, top_k, top_p,
                  rep_penalty, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)row_offset};
  hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, p);
  VIDIL_CHECK_LAUNCH("sample");
  return VIDIL_OK;
}
