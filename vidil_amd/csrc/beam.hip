// beam.hip — device-side beam search bookkeeping so a whole caption batch
// decodes without a host round trip per step.  Semantics follow HF
// transformers 4.15 (generation_utils.beam_search + BeamSearchScorer +
// MinLengthLogitsProcessor), the version models/med.py:7-8 of the reference
// names; the reference call site is models/blip.py:154-161.
//
//   logsoftmax_topk : one workgroup per image; log-softmax of its nb rows (one
//                     pass per row: online max / sum of exponentials),
//                     optional EOS ban, + beam score, sorted top-2nb over
//                     nb*V candidates (ties -> lower flat index).
//   beam_update     : one thread per image; BeamSearchScorer.process + the
//                     input_ids gather/append.
//   beam_finalize   : one thread per image; BeamSearchScorer.finalize.
#include <math.h>
#include "common.h"

namespace {

constexpr int MAXC = 8;   // 2*nb candidates, nb <= 4
constexpr int MAXNB = 4;
constexpr int MAXLEN = 64;

struct Cand {
  float s;
  int i;
};
__device__ __forceinline__ bool better(float s1, int i1, float s2, int i2) {
  return s1 > s2 || (s1 == s2 && i1 < i2);
}
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float s = __shfl_xor(c.s, o, 64);
    const int i = __shfl_xor(c.i, o, 64);
    if (better(s, i, c.s, c.i)) { c.s = s; c.i = i; }
  }
  return c;
}

__device__ float block_reduce_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ float block_reduce_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// RP: the repetition penalty of `generate(..., repetition_penalty != 1.0)` with beam search (reference: models/blip.py:161
// passes it to HF generate, whose RepetitionPenaltyLogitsProcessor runs FIRST in the processor list and, in beam search, on the
// LOG-PROBABILITIES: every token already in a row's sequence — prompt included — gets `s < 0 ? s * penalty : s / penalty`; oracle:
// oracle/beam_ref.py, pinned against the installed transformers by tests/test_beam_hf.py).  The row's history tokens are kept
// out of the raw-logit candidates (a penalty < 1 RAISES their score, so "best logits = best scores" does not hold for them)
// and are scored one per thread once the row's log-sum-exp is known.
template <int NB, bool RP>
__global__ __launch_bounds__(256) void lsm_topk_kernel(const float* __restrict__ logits,
                                                       const float* __restrict__ beam_scores, int NBL, int V,
                                                       int ban, float* __restrict__ out_s, int* __restrict__ out_i,
                                                       const int32_t* __restrict__ hist, int hist_len, int ld_hist,
                                                       float penalty) {
  constexpr int K = 2 * NB;
  __shared__ float red[4];
  __shared__ int hs_tok[MAXLEN];
  __shared__ float ls[256 * K];
  __shared__ int li[256 * K];
  __shared__ Cand wbest[4];
  const int b = blockIdx.x, tid = threadIdx.x;

  float ts[K];
  int ti[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { ts[j] = -INFINITY; ti[j] = 0x7fffffff; }

  const bool vec = (V & 3) == 0;   // rows are then 16-B aligned: 4 logits per load
  auto consider = [&](float c, int flat) {
    if (better(c, flat, ts[K - 1], ti[K - 1])) {
      ts[K - 1] = c; ti[K - 1] = flat;
#pragma unroll
      for (int j = K - 1; j > 0; --j) {
        if (better(ts[j], ti[j], ts[j - 1], ti[j - 1])) {
          const float s = ts[j]; ts[j] = ts[j - 1]; ts[j - 1] = s;
          const int x = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = x;
        }
      }
    }
  };
  // ONE pass over the logits per row (round 4; the round-1 form read every row three times: max, sum of exponentials,
  // candidates — 3.68 GB instead of 1.31 GB per launch at 10,752 x 30,524): every thread keeps an ONLINE softmax state
  // (running maximum m, sum of exp(x - m), rescaled when the maximum moves) and its K best RAW logits of the row; the
  // block then combines the states into the row's (max, log-sum-exp), and only the K survivors per thread are turned into
  // scores `(x - max) - lse + beam_score` and merged into the thread's candidates of the image.  log_softmax is monotone
  // in x, so a row's best scores are its best logits (ties in the rounded score are ordered by the final merge, which
  // compares (score, flat index) exactly as before).
  constexpr float L2E = 1.4426950408889634f;
  for (int beam = 0; beam < NBL; ++beam) {
    const float* row = logits + ((size_t)b * NBL + beam) * V;
    if (RP) {
      __syncthreads();                                   // (the previous row's readers are done)
      if (tid < hist_len) hs_tok[tid] = hist[((size_t)b * NB + beam) * ld_hist + tid];
      __syncthreads();
    }
    auto in_hist = [&](int t) {
      bool f = false;
      for (int h = 0; h < hist_len; ++h) f |= hs_tok[h] == t;
      return f;
    };
    float m = -INFINITY, sum = 0.f;
    float rs[K];
    int ri[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { rs[j] = -INFINITY; ri[j] = 0x7fffffff; }
    auto consider_raw = [&](float x, int i) {
      if (better(x, i, rs[K - 1], ri[K - 1])) {
        rs[K - 1] = x; ri[K - 1] = i;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
          if (better(rs[j], ri[j], rs[j - 1], ri[j - 1])) {
            const float t = rs[j]; rs[j] = rs[j - 1]; rs[j - 1] = t;
            const int x_ = ri[j]; ri[j] = ri[j - 1]; ri[j - 1] = x_;
          }
        }
      }
    };
    if (vec) {
      // Candidate filter (second half of round 4).  A thread sees ~120 logits of a row, so some lane of a wave has a new
      // personal best at nearly every element and the (divergent) sorted insertion ran for all of them: ~200 instructions
      // per 16-byte load, the kernel's bound (2.9 TB/s).  `thr` is a wave-uniform LOWER bound of the row's K-th best logit
      // — the K-th largest of the lanes' current best candidates, which are K distinct elements of the row — minus a margin
      // far wider than any pair of logits whose rounded scores could tie: an element below it cannot be among the row's K
      // best in the (score, index) order of the final merge, so skipping it changes no output (same digests, 459 -> 347 us;
      // the bound re-derived every 2 / 4 / 8 / 16 loads: 430 / 360 / 347 / 358 us; four loads per thread issued ahead of
      // their use on top of it: 374 us — slower, left out).
      float thr = -INFINITY;
      int it = 0;
      for (int i = tid * 4; i < V; i += 1024, ++it) {
        const f32x4 x = *(const f32x4*)(row + i);
        const float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
        if (mx > m) {                                   // (m = -inf at first: sum = 0 * 2^-inf = 0)
          sum *= __builtin_amdgcn_exp2f((m - mx) * L2E);
          m = mx;
        }
        const float ml = m * L2E;
        sum += (__builtin_amdgcn_exp2f(__builtin_fmaf(x[0], L2E, -ml)) + __builtin_amdgcn_exp2f(__builtin_fmaf(x[1], L2E, -ml))) +
               (__builtin_amdgcn_exp2f(__builtin_fmaf(x[2], L2E, -ml)) + __builtin_amdgcn_exp2f(__builtin_fmaf(x[3], L2E, -ml)));
        if (__builtin_amdgcn_ballot_w64(mx >= thr) != 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (i + e != ban && x[e] >= thr && !(RP && in_hist(i + e))) consider_raw(x[e], i + e);  // (the banned token counts in the softmax, never as a candidate)
        }
        // (the refresh shuffles across the wave, so it runs only while EVERY lane is still inside the loop — on the
        //  ragged last iteration of a vocabulary that is not a multiple of 1,024 some lanes have left, their registers are
        //  undefined to a shuffle, and a bound derived from them could exceed real logits)
        if ((it & 7) == 1 && __builtin_amdgcn_ballot_w64(true) == ~0ull) {
          float c = rs[0], t = -INFINITY;
#pragma unroll
          for (int r = 0; r < K; ++r) {
            t = wave_max(c);
            const unsigned long long holders = __builtin_amdgcn_ballot_w64(c == t);
            if (holders == 0) break;                     // (NaNs: leave the bound where it is)
            if ((int)(threadIdx.x & 63) == __builtin_ctzll(holders)) c = -INFINITY;   // one holder leaves per round
          }
          const float bound = t - 1e-5f * (fabsf(t) + 64.f);
          if (bound > thr) thr = bound;                  // (t = -inf while the wave has seen fewer than K elements)
        }
      }
    } else {
      for (int i = tid; i < V; i += 256) {
        const float x = row[i];
        if (x > m) {
          sum *= __builtin_amdgcn_exp2f((m - x) * L2E);
          m = x;
        }
        sum += __builtin_amdgcn_exp2f((x - m) * L2E);
        if (i != ban && !(RP && in_hist(i))) consider_raw(x, i);
      }
    }
    const float M = block_reduce_max(m, red);
    // (a thread that saw no element has m = -inf, sum = 0: contributes 0)
    sum = block_reduce_sum(m == -INFINITY ? 0.f : sum * __builtin_amdgcn_exp2f((m - M) * L2E), red);
    const float lse = logf(sum);
    const float bs = beam_scores[b * NB + beam];
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (ri[j] != 0x7fffffff) consider((rs[j] - M) - lse + bs, beam * V + ri[j]);
    if (RP && tid < hist_len) {
      const int t = hs_tok[tid];
      bool first = t >= 0 && t < V && t != ban;          // (scatter_ of equal values: a repeated token is penalised once)
      for (int h = 0; h < tid; ++h) first &= hs_tok[h] != t;
      if (first) {
        const float lp = (row[t] - M) - lse;
        consider((lp < 0.f ? lp * penalty : lp / penalty) + bs, beam * V + t);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) { ls[tid * K + j] = ts[j]; li[tid * K + j] = ti[j]; }
  __syncthreads();
  int head = 0;
  for (int round = 0; round < K; ++round) {
    Cand c;
    c.s = head < K ? ls[tid * K + head] : -INFINITY;
    c.i = head < K ? li[tid * K + head] : 0x7fffffff;
    c = wave_best(c);
    if ((tid & 63) == 0) wbest[tid >> 6] = c;
    __syncthreads();
    Cand g = wbest[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (better(wbest[w].s, wbest[w].i, g.s, g.i)) g = wbest[w];
    if (head < K && li[tid * K + head] == g.i) ++head;  // flat indices are unique
    if (tid == 0) { out_s[b * K + round] = g.s; out_i[b * K + round] = g.i; }
    __syncthreads();
  }
}

// ---- BeamHypotheses.add (double arithmetic: the reference does this on Python floats)
__device__ void hyp_add(const vidil_beam_state& st, int b, int nb, int max_len, const int32_t* toks, int len,
                        double sum_logprobs) {
  const double score = sum_logprobs / (double)len;
  int n = st.n_hyp[b];
  double* hs = st.hyp_score + (size_t)b * nb;
  int32_t* hl = st.hyp_len + (size_t)b * nb;
  int32_t* ht = st.hyp_tok + (size_t)b * nb * max_len;
  if (n < nb) {
    hs[n] = score; hl[n] = len;
    for (int t = 0; t < len; ++t) ht[(size_t)n * max_len + t] = toks[t];
    st.n_hyp[b] = n + 1;
    st.worst[b] = fmin(score, st.worst[b]);
    return;
  }
  if (!(score > st.worst[b])) return;
  // list is full: the new one is appended, then the lowest (score, position) is deleted.
  int lo = 0;
  for (int j = 1; j < nb; ++j)
    if (hs[j] < hs[lo]) lo = j;
  // (the appended entry has score > worst == hs[lo], so it is never the one deleted)
  for (int j = lo; j + 1 < nb; ++j) {
    hs[j] = hs[j + 1]; hl[j] = hl[j + 1];
    for (int t = 0; t < hl[j]; ++t) ht[(size_t)j * max_len + t] = ht[(size_t)(j + 1) * max_len + t];
  }
  hs[nb - 1] = score; hl[nb - 1] = len;
  for (int t = 0; t < len; ++t) ht[(size_t)(nb - 1) * max_len + t] = toks[t];
  double w = hs[0];
  for (int j = 1; j < nb; ++j) w = fmin(w, hs[j]);
  st.worst[b] = w;
}

__global__ void beam_update_kernel(const vidil_beam_state st, const float* __restrict__ cand_s,
                                   const int32_t* __restrict__ cand_i, int B, int nb, int V, int cur_len, int max_len,
                                   int eos_id, int pad_id) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int K = 2 * nb;
  float ns[MAXNB];
  int nt[MAXNB], nsrc[MAXNB];
  if (st.done[b]) {
    for (int j = 0; j < nb; ++j) { ns[j] = 0.f; nt[j] = pad_id; nsrc[j] = b * nb + j; }
  } else {
    int slot = 0;
    for (int rank = 0; rank < K && slot < nb; ++rank) {
      const int flat = cand_i[b * K + rank];
      const float sc = cand_s[b * K + rank];
      const int beam = flat / V, tok = flat - beam * V;
      const int src = b * nb + beam;
      if (tok == eos_id) {
        if (rank >= nb) continue;
        hyp_add(st, b, nb, max_len, st.seqs + (size_t)src * max_len, cur_len, (double)sc);
      } else {
        ns[slot] = sc; nt[slot] = tok; nsrc[slot] = src; ++slot;
      }
    }
    for (; slot < nb; ++slot) { ns[slot] = -1e9f; nt[slot] = pad_id; nsrc[slot] = b * nb; }  // unreachable
    if (st.n_hyp[b] >= nb) {
      const double cur_score = (double)cand_s[b * K] / (double)cur_len;
      if (st.worst[b] >= cur_score) st.done[b] = 1;
    }
  }
  for (int j = 0; j < nb; ++j) {
    const int row = b * nb + j;
    st.beam_scores[row] = ns[j];
    st.beam_idx[row] = nsrc[j];
    st.next_tok[row] = nt[j];
    const int32_t* src = st.seqs + (size_t)nsrc[j] * max_len;
    int32_t* dst = st.seqs_next + (size_t)row * max_len;
    for (int t = 0; t < cur_len; ++t) dst[t] = src[t];
    dst[cur_len] = nt[j];
  }
  if (st.done[b] && st.n_done != nullptr) atomicAdd(st.n_done, 1);
}

__global__ void beam_finalize_kernel(const vidil_beam_state st, int B, int nb, int cur_len, int max_len, int eos_id,
                                     int pad_id, int32_t* __restrict__ out_tok, int32_t* __restrict__ out_len,
                                     float* __restrict__ out_score) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (!st.done[b]) {
    for (int j = 0; j < nb; ++j) {
      const int row = b * nb + j;
      hyp_add(st, b, nb, max_len, st.seqs + (size_t)row * max_len, cur_len, (double)st.beam_scores[row]);
    }
  }
  // sorted(beams, key=score).pop(): the highest score, the LAST one among equals
  const int n = st.n_hyp[b];
  const double* hs = st.hyp_score + (size_t)b * nb;
  int best = 0;
  for (int j = 1; j < n; ++j)
    if (hs[j] >= hs[best]) best = j;
  const int len = st.hyp_len[(size_t)b * nb + best];
  const int32_t* ht = st.hyp_tok + ((size_t)b * nb + best) * max_len;
  for (int t = 0; t < max_len; ++t) out_tok[(size_t)b * max_len + t] = t < len ? ht[t] : pad_id;
  if (len < max_len) out_tok[(size_t)b * max_len + len] = eos_id;
  out_len[b] = len;
  out_score[b] = (float)hs[best];
}

}  // namespace

static int launch_lsm_topk(const float* logits, const float* beam_scores, int32_t B, int32_t nb, int32_t nbl, int32_t V,
                           int32_t ban_token, float* out_scores, int32_t* out_index, const int32_t* hist, int32_t hist_len,
                           int32_t ld_hist, float penalty, void* stream, const char* who) {
  VIDIL_REQUIRE(nbl >= 1 && nbl <= nb, "%s: beams_in_logits=%d must be in [1, num_beams=%d]", who, nbl, nb);
  VIDIL_REQUIRE(logits && beam_scores && out_scores && out_index, "%s: null pointer", who);
  VIDIL_REQUIRE(B > 0 && V > 0, "%s: bad shape", who);
  VIDIL_REQUIRE((long)nb * V < 0x7fffffffL, "%s: nb*V overflows int32", who);
  hipStream_t s = (hipStream_t)stream;
#define VIDIL_LSM(NB_)                                                                                                          \
  case NB_:                                                                                                                     \
    if (hist) hipLaunchKernelGGL((lsm_topk_kernel<NB_, true>), dim3(B), dim3(256), 0, s, logits, beam_scores, nbl, V, ban_token, \
                                 out_scores, out_index, hist, hist_len, ld_hist, penalty);                                      \
    else hipLaunchKernelGGL((lsm_topk_kernel<NB_, false>), dim3(B), dim3(256), 0, s, logits, beam_scores, nbl, V, ban_token,     \
                            out_scores, out_index, nullptr, 0, 0, 1.f);                                                         \
    break;
  switch (nb) {
    VIDIL_LSM(1) VIDIL_LSM(2) VIDIL_LSM(3) VIDIL_LSM(4)
    default:
      vidil_set_error("%s: num_beams=%d not supported (1..4)", who, nb);
      return VIDIL_EUNSUP;
  }
#undef VIDIL_LSM
  VIDIL_CHECK_LAUNCH(who);
  return VIDIL_OK;
}

extern "C" int vidil_logsoftmax_topk(const float* logits, const float* beam_scores, int32_t B, int32_t nb,
                                     int32_t beams_in_logits, int32_t V, int32_t ban_token, float* out_scores,
                                     int32_t* out_index, void* stream) {
  return launch_lsm_topk(logits, beam_scores, B, nb, beams_in_logits, V, ban_token, out_scores, out_index, nullptr, 0, 0, 1.f,
                         stream, "logsoftmax_topk");
}

extern "C" int vidil_logsoftmax_topk_penalty(const float* logits, const float* beam_scores, int32_t B, int32_t nb,
                                             int32_t beams_in_logits, int32_t V, int32_t ban_token, const int32_t* seqs,
                                             int32_t cur_len, int32_t ld_seqs, float penalty, float* out_scores,
                                             int32_t* out_index, void* stream) {
  VIDIL_REQUIRE(seqs != nullptr, "logsoftmax_topk_penalty: null sequences");
  VIDIL_REQUIRE(cur_len >= 1 && cur_len <= MAXLEN && ld_seqs >= cur_len,
                "logsoftmax_topk_penalty: cur_len=%d must be in [1, %d] and <= ld_seqs=%d", cur_len, MAXLEN, ld_seqs);
  VIDIL_REQUIRE(penalty > 0.f, "logsoftmax_topk_penalty: penalty must be a strictly positive float");
  return launch_lsm_topk(logits, beam_scores, B, nb, beams_in_logits, V, ban_token, out_scores, out_index, seqs, cur_len,
                         ld_seqs, penalty, stream, "logsoftmax_topk_penalty");
}

static int check_state(const vidil_beam_state* st, int nb, int max_len, const char* who) {
  VIDIL_REQUIRE(st != nullptr, "%s: null state", who);
  VIDIL_REQUIRE(st->seqs && st->seqs_next && st->beam_scores && st->beam_idx && st->next_tok && st->done && st->n_hyp &&
                    st->hyp_score && st->hyp_len && st->hyp_tok && st->worst,
                "%s: null pointer in state", who);
  VIDIL_REQUIRE(nb >= 1 && nb <= MAXNB, "%s: num_beams=%d not supported (1..4)", who, nb);
  VIDIL_REQUIRE(max_len >= 2 && max_len <= MAXLEN, "%s: max_len=%d not supported (2..64)", who, max_len);
  return VIDIL_OK;
}

extern "C" int vidil_beam_update(const vidil_beam_state* st, const float* cand_scores, const int32_t* cand_index,
                                 int32_t B, int32_t nb, int32_t V, int32_t cur_len, int32_t max_len, int32_t eos_id,
                                 int32_t pad_id, void* stream) {
  int rc = check_state(st, nb, max_len, "beam_update");
  if (rc) return rc;
  VIDIL_REQUIRE(cand_scores && cand_index && B > 0 && V > 0, "beam_update: bad args");
  VIDIL_REQUIRE(cur_len >= 1 && cur_len < max_len, "beam_update: cur_len=%d must be in [1,max_len)", cur_len);
  hipStream_t s = (hipStream_t)stream;
  if (st->n_done != nullptr) {
    hipError_t e = hipMemsetAsync(st->n_done, 0, sizeof(int32_t), s);
    if (e != hipSuccess) { vidil_set_error("beam_update: memset failed: %s", hipGetErrorString(e)); return VIDIL_ELAUNCH; }
  }
  hipLaunchKernelGGL(beam_update_kernel, dim3((B + 63) / 64), dim3(64), 0, s, *st, cand_scores, cand_index, B, nb, V,
                     cur_len, max_len, eos_id, pad_id);
  VIDIL_CHECK_LAUNCH("beam_update");
  return VIDIL_OK;
}

extern "C" int vidil_beam_finalize(const vidil_beam_state* st, int32_t B, int32_t nb, int32_t cur_len, int32_t max_len,
                                   int32_t eos_id, int32_t pad_id, int32_t* out_tokens, int32_t* out_len,
                                   float* out_score, void* stream) {
  int rc = check_state(st, nb, max_len, "beam_finalize");
  if (rc) return rc;
  VIDIL_REQUIRE(out_tokens && out_len && out_score && B > 0, "beam_finalize: bad args");
  VIDIL_REQUIRE(cur_len >= 1 && cur_len <= max_len, "beam_finalize: cur_len=%d", cur_len);
  hipLaunchKernelGGL(beam_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, *st, B, nb, cur_len,
                     max_len, eos_id, pad_id, out_tokens, out_len, out_score);
  VIDIL_CHECK_LAUNCH("beam_finalize");
  return VIDIL_OK;
}
