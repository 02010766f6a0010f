/*
 * vidil_hip.h — C ABI of libvidil_hip.so, the MI355X (gfx950) kernels behind
 * VidIL's frame-encoding hot path.
 *
 * The reference is pure Python on PyTorch (no FFI of its own); every entry
 * point below replaces the torch/cuDNN/cuBLAS op sequence that one reference
 * function issues.  The "replaces" line cites that function as file:line under
 * the reference tree (MikeWangWZHL/VidIL).
 *
 * Conventions
 *   - every function returns 0 on success, a negative VIDIL_E* code on error;
 *     vidil_last_error() returns a static, thread-local message for the last
 *     failure on the calling thread.
 *   - all pointers are DEVICE pointers unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream()
 *     .cuda_stream on the Python side).  Nothing here allocates, frees or
 *     synchronises; workspaces are caller-owned.
 *   - 16-bit operand type: every entry point that touches MFMA operands takes a
 *     `dtype` code, VIDIL_DT_F16 (IEEE half) or VIDIL_DT_BF16 (bfloat16); "T16"
 *     below means "that type".  Accumulation, LayerNorm statistics, softmax and
 *     the residual stream are f32 for both.
 *   - head_dim is 64 everywhere on this path (ViT-B/L, MED/BERT, CLIP B/32 and
 *     L/14 towers all use 64).
 */
#ifndef VIDIL_HIP_H
#define VIDIL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* operand types (the `dtype` arguments) */
#define VIDIL_DT_F16 0
#define VIDIL_DT_BF16 1
#define VIDIL_DT_FP8 2      /* OCP e4m3fn: GEMM A / W operands (and the tensors feeding them) of the fp8 tower mode */
/* Flag OR-ed onto the 16-bit type of an OUTPUT (vidil_layernorm dtype16, vidil_patchify_* dtype, vidil_attention /
 * vidil_beam_attention out_dtype): the rows are written as error-compensated GEMM operands [hi | lo | hi] — three planes
 * of the logical row width, hi = T16(x), lo = T16(x - hi) — exactly what vidil_split3_f32 makes of an f32 row.  Against
 * a weight [W_hi | W_hi | W_lo] one K-tripled GEMM then yields x·W to ~2^-21 relative instead of ~2^-11 (the "parity"
 * precision mode of vidil_amd: caption logits within 1e-3 of the fp32 reference). */
#define VIDIL_DT_SPLIT3 0x100
/* with VIDIL_DT_SPLIT3 (vidil_layernorm; vidil_attention_f32's out_mode 3): rows laid out as three planes, but only planes
 * hi | lo are written — for a consumer that is a split_k GEMM in the K-loop form (vidil_gemm_split_k_in_loop), which reads
 * planes 0 / 1 of its A rows only (ABI 10, round 5) */
#define VIDIL_DT_SPLIT2 0x200

#define VIDIL_OK 0
#define VIDIL_EINVAL (-1)   /* bad argument (shape, alignment, null pointer)  */
#define VIDIL_ELAUNCH (-2)  /* hipLaunchKernel / hipFuncSetAttribute failed   */
#define VIDIL_EUNSUP (-3)   /* shape outside what the kernels were built for  */

const char* vidil_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int vidil_abi_version(void);
/* Number of symbols a loader must find; used by the not-gpu load test. */
int vidil_num_entry_points(void);

/* ------------------------------------------------------------------------ */
/* GEMM  C[M,N] = A[M,K] · W[N,K]^T  (+bias) with a fused epilogue.           */
/* A and W are T16 row-major with K contiguous (torch nn.Linear layout).      */
/* K must be a multiple of 64.  M, N arbitrary (>0).                          */
/* ------------------------------------------------------------------------ */
enum {
  VIDIL_EPI_F16 = 0,   /* out T16 [M,ldo]   = act(acc + bias)                              */
  VIDIL_EPI_F32 = 1,   /* out f32 [M,ldo]   = act(acc + bias) + resid (resid may alias out)*/
  VIDIL_EPI_HEADS = 2, /* scatter into per-head Q / K / V^T buffers (see below)            */
  VIDIL_EPI_PATCH = 3, /* out f32 row (m + m/tpi + 1) = acc + bias + pos[(m%tpi)+1]        */
  VIDIL_EPI_ARENA = 4, /* Q rows + K / V rows appended to a beam-search KV arena (below)   */
  VIDIL_EPI_F8 = 5     /* out fp8 (e4m3) [M,ldo] = act(acc * w_scale + bias): the fc1 -> fc2 hand-over of the fp8 mode */
};
enum { VIDIL_ACT_NONE = 0, VIDIL_ACT_GELU_ERF = 1, VIDIL_ACT_QUICK_GELU = 2 };

typedef struct vidil_gemm_args {
  const void* A;      /* T16 [M,K], row stride lda                             */
  const void* W;      /* T16 [N,K]                                             */
  const float* bias;  /* f32 [N] or NULL                                       */
  int32_t M, N, K;
  int32_t lda;        /* A row stride in elements; 0 = K; multiple of 8        */
  int32_t epi;        /* VIDIL_EPI_*                                           */
  int32_t act;        /* VIDIL_ACT_* (EPI_F16 / EPI_F32 only)                  */
  void* out;          /* EPI_F16: T16, EPI_F32/PATCH: f32, row stride ldo      */
  int32_t ldo;
  const float* resid; /* EPI_F32: f32 [M,ldo] added after act, or NULL         */
  /* EPI_HEADS: column n -> part = part0 + n/(H*64), h = (n%(H*64))/64, d=n%64;
   * row m -> b = m/T, t = m%T.
   *   part 0: Q [b][h][t][64]            T16, value * q_scale, row capacity Tq_cap
   *   part 1: K [b][h][t_off+t][64]      T16, row capacity Tk_cap
   *   part 2: VT[b][h][d][vt(t_off+t)]   T16, row stride NP (multiple of 16); keys of
   *           every 16-key block are stored in the order 0-3, 8-11, 4-7, 12-15:
   *           vt(t) = t ^ 12 when bits 2 and 3 of t differ, else t (see vidil_attention)
   *   NP == 0: part 2 is stored ROW-MAJOR instead, V [b][h][t_off+t][64] in `vt` with row capacity Tk_cap
   *           (16-byte stores like K; consumed by vidil_attention's staged kernel, which transposes in LDS) */
  void* q;
  void* k;
  void* vt;
  int32_t T, H, part0, t_off, Tq_cap, Tk_cap, NP;
  float q_scale;
  /* EPI_ARENA: same column / row decomposition as EPI_HEADS (T, H, part0, t_off, q_scale), but the
   * destinations are row-major with all heads of a token contiguous (no transposition, no per-head
   * scatter — the layout vidil_beam_attention reads):
   *   part 0: Q     [m][H*64]                                   T16, value * q_scale
   *   part 1: K     [t_off+t][b*slot_stride][H*64]  (arena `k`)  T16, arena_rows slots per position
   *   part 2: V     [t_off+t][b*slot_stride][H*64]  (arena `vt`) T16 (NOT transposed)
   * decode step: T = 1, slot_stride = 1 (row m appends position t_off of slot m);
   * shared prompt pass: T = P, slot_stride = nb (image b's prompt lives in slot b*nb). */
  int32_t arena_rows, slot_stride;
  /* EPI_PATCH */
  const float* pos;   /* f32 [(tpi+1), N]                                      */
  int32_t tpi;        /* patches per image                                     */
  /* EPI_HEADS, parts 1 and 2: non-zero = FRAGMENT-TILED K and V (NP ignored, Tk_cap a multiple of 32): every
   * (b, h) owns Tk_cap/32 tiles of 32 keys x 64 dims (2048 T16) in the operand order of the direct attention kernel,
   *   K tile [c/8][key%32][c%8];  V tile [key%32/16][d/32][g%2][d%32][(g/2)*4 + key%4], g = (key%16)/4,
   * so each of that kernel's wave loads is one contiguous KiB.  Consumer: vidil_attention(kv_tiled = 1). */
  int32_t kv_tiled;
  int32_t dtype;      /* VIDIL_DT_*: type of A, W and of every 16-bit output (out, q, k, vt)   */
  /* ---- LayerNorm folded into the GEMMs of a pre-LN block (models/vit.py:107-110 `x + attn(norm1(x))`,
   * `x + mlp(norm2(x))`; HF CLIPEncoderLayer) -----------------------------------------------------
   * producer (EPI_F32, the residual GEMM that writes the stream x): out16 != NULL additionally stores
   *   T16(x) [M, ldo16] — the RAW stream in the operand type — and, per row and per 64 output columns, the
   *   partial (sum, sum of squares) of the f32 values into ln_stats_out f32 [M][N/64][2] (N % 64 == 0);
   * consumer (EPI_F16 / EPI_HEADS / EPI_ARENA (round 6) with ln_fold != 0; needs K == the LayerNorm width D): A is that raw T16(x),
   *   W is W' = T16(gamma (.) W) (gamma scales the K axis), bias is b' = b + W·beta, ln_colsum[n] = sum_k W'[n][k],
   *   ln_stats = the producer's partials [M][K/64][2]; the kernel combines them into mean_m, rstd_m and applies
   *   y[m][n] = rstd_m * (acc[m][n] - mean_m * ln_colsum[n]) + b'[n]  in its epilogue (then act / the
   *   per-head scatter), which equals LayerNorm(x)·W^T + b with the rounding point moved from LN(x) to x.
   * Both run on the 256x256 kernel whatever M is (results never depend on the batch size). */
  void* out16;
  int32_t ldo16;
  int32_t ln_fold;
  const float* ln_colsum; /* f32 [N] */
  float ln_eps;
  float* ln_stats_out;    /* producer: f32 [M][N/64][2] */
  const float* ln_stats;  /* consumer: f32 [M][K/64][2] */
  /* ---- fp8 tower mode (dtype == VIDIL_DT_FP8; BASELINE config 5 "fp8 MFMA ViT path") ----------------------------
   * A and W are OCP e4m3 bytes (K % 128 == 0), multiplied by v_mfma_scale_f32_32x32x64_f8f6f4 (block scales fixed at
   * 2^0: plain fp8 products at twice the f16 MFMA rate, f32 accumulate).  W is stored as W / w_scale[n] (per output
   * column, chosen by the host so that the row's largest weight sits well inside e4m3's range); the epilogue
   * multiplies the accumulator by w_scale[n] before the bias.  dtype16 names the 16-bit type of the 16-bit outputs
   * (EPI_HEADS q / k / vt); EPI_F8 writes fp8; EPI_F32 / EPI_PATCH as for the 16-bit types.  The 256x256 kernel only. */
  const float* w_scale;   /* f32 [N] or NULL (= 1) */
  int32_t dtype16;        /* VIDIL_DT_F16 / VIDIL_DT_BF16 */
  /* LayerNorm of the RESIDUAL (post-LN stacks, models/med.py:236-239,306-317: h = LN(x + dense(...)) feeds the next
   * dense AND is the next residual).  Non-NULL (both, f32 [N]) with epi F32, resid, ln_stats and ln_stats_out: `resid`
   * holds the previous block's RAW sum u; the epilogue adds ((u - mean) * rstd) * rln_gamma[n] + rln_beta[n] instead of u,
   * mean / rstd of row m from the partials `ln_stats` ([M][N/64][2], written by the GEMM that produced u; ln_eps).
   * ln_fold stays 0 (A is not normalised here).  N % 64 == 0, N <= 1024. */
  const float* rln_gamma;
  const float* rln_beta;
  /* EPI_F32 with out16 (ABI 10, round 5; no ln_stats_out / rln): non-zero = the 16-bit copy is written as ERROR-COMPENSATED
   * operand rows [hi | lo | hi] (VIDIL_DT_SPLIT3's layout: three planes of ldo16 / 3 >= N columns; hi = T16(v), lo =
   * T16(v - hi)) of v = act(acc + bias) + resid — what vidil_split3_f32 would make of the f32 output, from the same f32
   * values — and `out` may then be NULL: the f32 rows are not written.  The parity precision mode's fc1 -> fc2 hand-over.
   * 2 = write planes hi | lo only (plane 2, a copy of plane 0, is left untouched): for a consumer that is a split_k launch in
   * the K-loop form (vidil_gemm_split_k_in_loop() != 0 and an f32 / per-head (T >= 8) / patch epilogue), which reads planes 0 / 1. */
  int32_t out16_split3;
  /* ERROR-COMPENSATED operands (ABI 10, round 5; the parity precision mode): non-zero = A rows are [x_hi | x_lo | x_hi]
   * (three planes of Kl = K / 3 columns: what VIDIL_DT_SPLIT3 outputs look like) and W rows are [W_hi | W_hi | W_lo], so
   * that the plain product over K = 3 Kl columns IS x_hi.W_hi + x_lo.W_hi + x_hi.W_lo (rounds 3-4 ran it as such).  With
   * the flag set the library may compute the same three products INSIDE one K loop over Kl (x_hi / W_hi tiles fetched once
   * instead of twice: 2/3 of the operand traffic per MFMA; f32 sums in a different order) — it does so for the f32, per-head
   * (T >= 8) and patch epilogues at EVERY problem size and runs the plain K = 3 Kl product for the others, so a row's
   * result never depends on the batch around it.  K % 96 == 0.
   * 2 (ABI 12, round 6) = as 1, and the caller states that ONLY planes 0 / 1 of the A rows were written (a producer run with
   * VIDIL_DT_SPLIT2 / out16_split3 = 2 / attention out_mode 3): the launch must take the K-loop form; when this call's
   * epilogue, alignment or shape does not qualify for it the library returns VIDIL_EINVAL instead of running the plain
   * K = 3 Kl product over an unwritten plane. */
  int32_t split_k;
} vidil_gemm_args;

/* replaces: nn.Linear calls of models/vit.py:35-41,72,84; models/med.py:153-171,
 * 236,301,314,512,534; timm PatchEmbed conv (models/vit.py:144-145,182) as an
 * im2col-free GEMM; HF CLIP q/k/v/out/fc1/fc2/projection Linears. */
int vidil_gemm(const vidil_gemm_args* args, void* stream);
/* 1 when split_k launches with an eligible epilogue take the K-loop form in this process (0: $VIDIL_GEMM_C3=0 — every split_k
 * launch is the plain K = 3 Kl product and reads all three planes of its A rows). */
int vidil_gemm_split_k_in_loop(void);
/* (ABI 12) Per CALL: 1 when vidil_gemm would run `args` (split_k != 0) in the K-loop form — planes 0 / 1 of the A rows are
 * all it reads, so their producer may be run with VIDIL_DT_SPLIT2 / out16_split3 = 2 / out_mode 3 —, 0 when it would run the plain
 * K = 3 Kl product over all three planes (split_k == 2 is then refused by vidil_gemm), < 0 for invalid arguments.  Launches
 * nothing.  Hosts size their producers' `planes` with it instead of assuming vidil_gemm_split_k_in_loop() covers every call. */
int vidil_gemm_split_k_serves(const vidil_gemm_args* args);
/* Name of the kernel instantiation vidil_gemm would launch for `args` (the spelling rocprofv3 prints, e.g.
 * "gemm256_kernel<f16, 1, 0>"), written NUL-terminated into buf[0..n).  For profilers / bench.py. */
int vidil_gemm_kernel_name(const vidil_gemm_args* args, char* buf_host, int32_t n);

/* ------------------------------------------------------------------------ */
/* LayerNorm over the last dim.  x f32 rows of length D at stride x_stride    */
/* (elements); writes T16 (dtype16; VIDIL_DT_FP8 too: the fp8 mode's GEMM operand;  */
/* T16 | VIDIL_DT_SPLIT3: rows [hi | lo | hi], 3D wide) and/or f32 outputs (either   */
/* may be NULL), dense.                                                            */
/* D must be a multiple of 64 and <= 1024... (768, 512, 1024 on this path)    */
/* replaces: nn.LayerNorm at models/vit.py:108-109,192 (eps 1e-6),            */
/* models/med.py:92,238,316,514 (eps 1e-12), HF CLIP layer norms (eps 1e-5).  */
/* ------------------------------------------------------------------------ */
int vidil_layernorm(const float* x, int64_t x_stride, const float* gamma,
                    const float* beta, float eps, int32_t M, int32_t D,
                    void* out16, int32_t dtype16, float* out_f32, void* stream);

/* ------------------------------------------------------------------------ */
/* Attention for short sequences (Nk <= 768): softmax(Q K^T [+mask]) V.       */
/* Q  T16 [Bq][H][Tq_cap][64] (already scaled by 1/sqrt(64)),                 */
/* K  T16 [Bk][H][Tk_cap][64], VT T16 [Bk][H][64][NP] with NP % 16 == 0 and    */
/* the key axis of every 16-key block permuted to 0-3, 8-11, 4-7, 12-15 (the   */
/* order the transposed-score MFMA layout consumes; written by EPI_HEADS).     */
/* NP == 0: `vt` holds V ROW-MAJOR [Bk][H][Tk_cap][64] (like K) and is          */
/* transposed while it is staged into LDS — allowed when every work unit has   */
/* more than 32 query rows (the encoder self-attention of the ViT / CLIP       */
/* towers); the short-query kernels read V^T fragments straight from memory.   */
/* The towers' own shape (row-major V, one query batch per K/V batch, 129..224 */
/* rows, 193..224 keys, no kv_len / causal, plain 16-bit output rows, 16-byte   */
/* aligned operands) runs on a persistent, LDS-DMA-streamed form of the staged */
/* kernel with bit-identical results (csrc/attention.hip: attn_stream_kernel). */
/* kv_tiled != 0: `k` and `vt` are FRAGMENT-TILED (vidil_gemm_args.kv_tiled;    */
/* [Bk][H][Tk_cap/32][2048], NP ignored, Tk_cap % 32 == 0) — allowed when every */
/* work unit has at most 32 query rows (the cross-attention of the caption      */
/* decoder, which re-reads the image K/V from HBM on every decode step).        */
/* Which key/value batch a query batch b reads — three forms, all of which let */
/* every query that shares a K/V (the captions of a frame, the beams of an     */
/* image) be served by ONE staging of that K/V:                               */
/*   group_start != NULL : kv batch j serves query batches                    */
/*                         group_start[j] .. group_start[j+1]-1 (n_kv batches, */
/*                         at most max_group query batches each);             */
/*   kv_index != NULL    : query batch b reads kv batch kv_index[b];           */
/*   otherwise           : kv batch j serves query batches j*kv_group ..      */
/*                         (j+1)*kv_group-1.                                  */
/* Keys >= kv_len[b] (or >= Nk when kv_len==NULL) are excluded; causal!=0     */
/* additionally excludes key > q + causal_off.                                */
/* out row (b*Nq + q), column h*64+d, row stride ldo (multiple of 8), in      */
/* out_dtype: the operand type `dtype`, or VIDIL_DT_FP8 (staged kernel only:  */
/* the fp8 tower mode feeds the attention output straight to the proj GEMM),  */
/* or dtype | VIDIL_DT_SPLIT3: row = three planes [hi | lo | hi] of ldo/3     */
/* columns each (ldo a multiple of 24, ldo/3 >= H*64).                        */
/* replaces: models/vit.py:75-83; models/med.py:178-220 (self, cross, cached);*/
/* HF CLIPAttention.                                                          */
/* ------------------------------------------------------------------------ */
int vidil_attention(const void* q, const void* k, const void* vt, void* out,
                    const int32_t* kv_len, const int32_t* kv_index,
                    const int32_t* group_start, int32_t n_kv, int32_t max_group,
                    int32_t Bq, int32_t H, int32_t Nq,
                    int32_t Nk, int32_t Tq_cap, int32_t Tk_cap, int32_t NP,
                    int32_t kv_group, int32_t causal, int32_t causal_off,
                    int32_t ldo, int32_t kv_tiled, int32_t dtype, int32_t out_dtype,
                    void* stream);

/* ------------------------------------------------------------------------ */
/* Attention in f32 — the attention of the "parity" precision mode (round 4). */
/* softmax(q k^T * scale) v per head (head_dim 64) in plain f32 arithmetic on  */
/* f32 Q / K / V that are read IN PLACE from the row-major outputs of the      */
/* projection GEMMs: element (row, head h, d) of an operand lives at           */
/* base + row * ld + off + h*64 + d (ld, off multiples of 4; no per-head       */
/* scatter).  Every GEMM operand of that mode is carried to ~2^-21; the 16-bit */
/* Q / K / V of vidil_attention were what was left of its error (2.4e-4 of the */
/* logit scale).  A precision mode, not a throughput path (~1/16 of the MFMA   */
/* kernels' arithmetic rate).                                                  */
/*  dense form (anc == NULL): query batch b has rows b*Nq .. b*Nq+Nq-1 of q;   */
/*   kv batch j has rows j*kv_rows .. j*kv_rows+Nk-1 of k / v; query batches   */
/*   map to kv batches by kv_group / kv_index / group_start (+ n_kv, max_group)*/
/*   and keys are masked by kv_len / causal / causal_off exactly as in         */
/*   vidil_attention.                                                          */
/*  arena form (anc != NULL, Nq == 1): key j of query row b is row             */
/*   j*arena_rows + anc[b*anc_ld + j] of k / v (the beam search's append-only  */
/*   KV arena in f32, vidil_beam_ancestry's table); Nk keys.                   */
/*  out row (b*Nq + t), column h*64 + d, row stride ldo: out_mode 0 = f32;     */
/*   out_mode 3 = as 2 with planes hi | lo only (VIDIL_DT_SPLIT2);             */
/*   out_mode 2 = dtype16 rows as three planes [hi | lo | hi] of ldo/3 columns */
/*   (what VIDIL_DT_SPLIT3 outputs look like: the next compensated GEMM's A).  */
/*  arith (ABI 10, round 5): 0 = plain f32 arithmetic (above); 1 = SPLIT-     */
/*   OPERAND form on the 16-bit matrix instruction: every operand of the two   */
/*   contractions as hi + lo of dtype16 (22 significant bits for f16), three    */
/*   MFMAs per contraction (a_hi.b_hi + a_lo.b_hi + a_hi.b_lo, f32 sums), f32  */
/*   softmax — ~1e-6 of the logit scale from f32 arithmetic at 1/5 of its cost. */
/*   Dense forms only (the arena form stays f32).  With arith == 1, kv16 != 0:  */
/*   k and v are NOT f32 rows but the 16-bit FRAGMENT TILES vidil_gemm writes   */
/*   (heads epilogue, kv_tiled; [n_kv][H][kv_rows/32][2048] of dtype16, kv_rows */
/*   a multiple of 32): the decode steps' cross-attention, at most 32 query rows */
/*   per unit — Q (f32, in place) and the probabilities are split, K / V are    */
/*   what the tiles hold (S = K.Q_lo + K.Q_hi, O = V.P_lo + V.P_hi).            */
/* replaces (in that mode): models/vit.py:75-83; models/med.py:178-220; HF     */
/* CLIPAttention.                                                              */
/* ------------------------------------------------------------------------ */
typedef struct {
  const float* q;
  const float* k;
  const float* v;
  void* out;
  int64_t ldq, ldk, ldv, ldo;
  int32_t q_off, k_off, v_off;
  int32_t out_mode, dtype16;
  int32_t Bq, H, Nq, Nk, kv_rows;
  int32_t kv_group;
  const int32_t* kv_index;
  const int32_t* group_start;
  int32_t n_kv, max_group;
  const int32_t* kv_len;
  int32_t causal, causal_off;
  const int32_t* anc;
  int32_t anc_ld, arena_rows;
  float scale;
  int32_t arith;   /* 0: f32 arithmetic; 1: split-operand 16-bit MFMA (ABI 10, see above) */
  int32_t kv16;    /* arith 1 only: k / v are dtype16 fragment tiles [n_kv][H][kv_rows/32][2048] */
} vidil_attn_f32_args;
int vidil_attention_f32(const vidil_attn_f32_args* args, void* stream);

/* ------------------------------------------------------------------------ */
/* One separable pass of Pillow's antialiased resize on 8-bit interleaved RGB */
/* frames (ImagingResampleHorizontal_8bpc / Vertical_8bpc of Pillow's          */
/* Resample.c): every output byte = clip8((2^21 + sum_i px_i * k_i) >> 22)     */
/* with 22-bit fixed-point weights — integer arithmetic, bit-exact with PIL.   */
/*   bounds i32 [n_out][2] = (first source index, tap count) per output index, */
/*   coeffs i32 [n_out][ksize] (device pointers; built by the host from        */
/*   precompute_coeffs + normalize_coeffs_8bpc, see vidil_amd/preprocess.py).  */
/*   vertical == 0: src u8 [B][in_h][in_w][3] -> dst [B][out_h][out_w][3],     */
/*     dst row y reads src row src_row0 + y (n_out = out_w);                   */
/*   vertical != 0: src [B][in_h][out_w][3] -> dst [B][out_h][out_w][3]        */
/*     (n_out = out_h, in_w == out_w, src_row0 == 0).                          */
/* replaces: transforms.Resize((S,S), BICUBIC) on PIL frames                   */
/* (run_video_CapFilt.py:128-134) and HF CLIPProcessor's shortest-edge resize  */
/* + centre crop (run_visual_tokenization.py:138-142), both PIL Image.resize.  */
/* ------------------------------------------------------------------------ */
int vidil_resample_u8(const uint8_t* src, uint8_t* dst, int32_t B, int32_t in_h,
                      int32_t in_w, int32_t out_h, int32_t out_w,
                      int32_t vertical, const int32_t* bounds,
                      const int32_t* coeffs, int32_t ksize, int32_t src_row0,
                      void* stream);

/* ------------------------------------------------------------------------ */
/* Frame -> patch rows (im2col for stride==kernel conv), fused with dtype     */
/* conversion.  out T16 [B*(S/ps)^2, 3*ps*ps], column = c*ps*ps + py*ps + px  */
/* (the flattening of a conv weight [N,3,ps,ps]); rows are zero padded to a   */
/* multiple of 64 columns.  dtype | VIDIL_DT_SPLIT3: rows [hi | lo | hi] of   */
/* the f32 pixel values, three times that width.                              */
/* replaces: timm PatchEmbed / HF CLIPVisionEmbeddings conv input read.       */
/* ------------------------------------------------------------------------ */
int vidil_patchify_f32(const float* img /*[B,3,S,S]*/, void* out, int32_t B,
                       int32_t S, int32_t ps, int32_t dtype, void* stream);
/* uint8 HWC frames, fused (x/255 - mean[c]) / std[c]:                        */
/* replaces run_video_CapFilt.py:128-137 (ToTensor+Normalize) and HF          */
/* CLIPImageProcessor rescale+normalize for frames already at S x S.          */
int vidil_patchify_u8(const uint8_t* img /*[B,S,S,3]*/, void* out, int32_t B,
                      int32_t S, int32_t ps, const float* mean3_host,
                      const float* std3_host, int32_t dtype, void* stream);
/* x[b*T + 0, :] = cls[:] + pos[0, :]   (models/vit.py:184-187)               */
int vidil_set_cls_row(float* x, const float* cls, const float* pos0, int32_t B,
                      int32_t T, int32_t D, void* stream);

/* ------------------------------------------------------------------------ */
/* Token embedding: out[m,:] = word[ids[m],:] + pos[pos_off + m%T,:]  (f32)   */
/* replaces models/med.py:85-91 and HF CLIPTextEmbeddings.                    */
/* ------------------------------------------------------------------------ */
int vidil_embed_tokens(const int32_t* ids, const float* word, const float* pos,
                       float* out, int32_t M, int32_t T, int32_t pos_off,
                       int32_t D, int32_t vocab, void* stream);

/* Error-compensated operand rows for a GEMM with K tripled (the "precise LM head", see vidil_amd/med.py):  */
/* out T16 [M, 3D] = [hi | lo | hi] with hi = T16(x), lo = T16(x - hi): against a weight [N, 3D] =           */
/* [W_hi | W_hi | W_lo] one GEMM yields x_hi·W_hi + x_lo·W_hi + x_hi·W_lo, i.e. x·W to ~2^-20 relative.      */
int vidil_split3_f32(const float* x, void* out16, int32_t M, int32_t D,
                     int32_t dtype, void* stream);

/* out[i,:] = x[idx[i],:]   (f32 rows of length D)                            */
int vidil_gather_rows_f32(const float* x, const int32_t* idx, float* out,
                          int32_t n, int32_t D, void* stream);
/* x[i,:] /= ||x[i,:]||_2   (in place, f32)   HF CLIP embeds normalisation    */
int vidil_l2_normalize_rows(float* x, int32_t n, int32_t D, void* stream);

/* ------------------------------------------------------------------------ */
/* Beam-search step on device (HF transformers 4.15 semantics, the version    */
/* models/med.py:7-8 names; call site models/blip.py:154-161).                */
/* ------------------------------------------------------------------------ */
/* per image b (rows b*nbl .. b*nbl+nbl-1 of logits [B*nbl,V]):                */
/*   lp = log_softmax(logits[row]);  if ban_token>=0: lp[ban_token] = -inf;   */
/*   cand = lp + beam_scores[b*nb+beam];  top (2*nb) over the candidates,     */
/*   sorted descending, ties -> lower flat index first.                       */
/* beams_in_logits (nbl) is nb normally; 1 on the first step, where all beams */
/* of an image are still identical and only beam 0 (score 0; the others carry */
/* -1e9 and can never reach the top 2nb) was run through the decoder.         */
/* out_scores f32 [B, 2nb], out_index i32 [B, 2nb] (flat index beam*V+tok).   */
int vidil_logsoftmax_topk(const float* logits, const float* beam_scores,
                          int32_t B, int32_t nb, int32_t beams_in_logits,
                          int32_t V, int32_t ban_token, float* out_scores,
                          int32_t* out_index, void* stream);

/* The same step under `generate(..., repetition_penalty = p != 1.0)` (reference:  */
/* models/blip.py:127,161 hands its argument to HF generate; the captioning call   */
/* site run_video_CapFilt.py:101 leaves it at 1.0): HF's                           */
/* RepetitionPenaltyLogitsProcessor runs first in the processor list and, in beam  */
/* search, on the log-probabilities — for every token t among the first cur_len    */
/* ids of the row's sequence seqs[(b*nb+beam)*ld_seqs ..] (prompt included):       */
/*   lp[t] = lp[t] < 0 ? lp[t] * penalty : lp[t] / penalty   (f32, once per token) */
/* before the ban (MinLength) and the beam score are applied.  cur_len <= 64.      */
int vidil_logsoftmax_topk_penalty(const float* logits, const float* beam_scores,
                                  int32_t B, int32_t nb, int32_t beams_in_logits,
                                  int32_t V, int32_t ban_token, const int32_t* seqs,
                                  int32_t cur_len, int32_t ld_seqs, float penalty,
                                  float* out_scores, int32_t* out_index, void* stream);

typedef struct vidil_beam_state {
  int32_t* seqs;        /* [B*nb, max_len] token ids (current beams)          */
  int32_t* seqs_next;   /* [B*nb, max_len] scratch, swapped by the caller     */
  float* beam_scores;   /* [B*nb]                                             */
  int32_t* beam_idx;    /* [B*nb] out: global source row of each new beam     */
  int32_t* next_tok;    /* [B*nb] out: token appended to each new beam        */
  int32_t* done;        /* [B]                                                */
  int32_t* n_hyp;       /* [B]                                                */
  double* hyp_score;    /* [B, nb]  (Python-float arithmetic in the reference) */
  int32_t* hyp_len;     /* [B, nb]                                            */
  int32_t* hyp_tok;     /* [B, nb, max_len]                                   */
  double* worst;        /* [B]  worst kept hypothesis score (init 1e9)        */
  int32_t* n_done;      /* [1]  number of finished images after this step     */
} vidil_beam_state;

/* BeamSearchScorer.process: walk the 2*nb candidates, bank EOS hypotheses,   */
/* fill the next beams, update done flags; then append tokens into seqs_next. */
int vidil_beam_update(const vidil_beam_state* st, const float* cand_scores,
                      const int32_t* cand_index, int32_t B, int32_t nb,
                      int32_t V, int32_t cur_len, int32_t max_len,
                      int32_t eos_id, int32_t pad_id, void* stream);
/* BeamSearchScorer.finalize: out_tokens i32 [B,max_len] = best hypothesis,    */
/* then eos_id if it fits, then pad_id; out_len i32 [B] (EOS not counted).    */
int vidil_beam_finalize(const vidil_beam_state* st, int32_t B, int32_t nb,
                        int32_t cur_len, int32_t max_len, int32_t eos_id,
                        int32_t pad_id, int32_t* out_tokens, int32_t* out_len,
                        float* out_score, void* stream);

/* ------------------------------------------------------------------------ */
/* Beam-search KV arena: the same _reorder_cache semantics (models/med.py:    */
/* 951-955) WITHOUT moving the cache.  Keys / values of every layer live in   */
/* an append-only arena [position][slot][H*64] (slot = the beam row that      */
/* produced them, written by EPI_ARENA); what is reordered each step is a     */
/* small ancestry table anc i32 [rows][Tcap]: anc[r][t] = the slot holding    */
/* position t of the sequence that beam row r currently continues.            */
/*   vidil_beam_ancestry: dst[r][t] = src[beam_idx[r]][t] for t < cur_pos,    */
/*                        dst[r][cur_pos] = r  (where row r's next K/V go).   */
/*   vidil_beam_attention: one query token per beam row r (q T16 [rows][H*64],*/
/*     pre-scaled) over positions 0..n_keys-1 of its ancestry:                */
/*     out[r][h*64+d] = softmax_t(q_rh . K[t][anc[r][t]][h]) V[t][anc[r][t]][h]*/
/*     f32 scores / softmax / accumulation, T16 output (row stride ldo;       */
/*     out_dtype = dtype, or dtype | VIDIL_DT_SPLIT3: planes ldo/3 apart).    */
/*     n_keys <= 64.  Replaces the cached self-attention of models/med.py:    */
/*     178-220 on decode steps (past_key_values + _reorder_cache).            */
/* ------------------------------------------------------------------------ */
int vidil_beam_ancestry(const int32_t* anc_src, int32_t* anc_dst,
                        const int32_t* beam_idx, int32_t rows, int32_t Tcap,
                        int32_t cur_pos, void* stream);
int vidil_beam_attention(const void* q, const void* k_arena, const void* v_arena,
                         const int32_t* anc, void* out, int32_t rows, int32_t H,
                         int32_t n_keys, int32_t arena_rows, int32_t Tcap,
                         int32_t ldo, int32_t dtype, int32_t out_dtype,
                         void* stream);

/* ------------------------------------------------------------------------ */
/* One nucleus-sampling step (HF transformers 4.15 sample() as configured by  */
/* models/blip.py:140-151: do_sample, top_p, repetition_penalty 1.1, BertConfig */
/* default top_k 50, min/max length).  Per unfinished row b of logits f32 [B,V]:*/
/*   repetition penalty on every distinct token of seqs[b][0..cur_len) (s<0 ?  */
/*   s*p : s/p); cur_len < min_length => s[eos] = -inf; keep the top_k scores  */
/*   (ties with the k-th kept); nucleus cut at top_p in descending order; draw */
/*   from the softmax of the survivors.  The draw is this library's contract   */
/*   (torch.multinomial's generator stream cannot be reproduced): u =          */
/*   Philox4x32-10(key = seed, counter = (row_offset + b, step, 0, 0))[0] >> 8 */
/*   scaled to [0,1), inverse CDF over candidates ordered (score desc, id asc). */
/* Writes next_tok[b] and seqs[b][cur_len] (pad for finished rows), sets       */
/* done[b] / increments n_done when eos is drawn.  V*4 B must fit 150 KB LDS.  */
/* ------------------------------------------------------------------------ */
int vidil_sample_top_k_top_p(const float* logits, int32_t* seqs, int32_t* done,
                             int32_t* n_done, int32_t* next_tok, int32_t B,
                             int32_t V, int32_t max_len, int32_t cur_len,
                             int32_t min_length, int32_t eos_id, int32_t pad_id,
                             int32_t top_k, float top_p, float rep_penalty,
                             uint64_t seed, int32_t step, int32_t row_offset,
                             void* stream);

/* ------------------------------------------------------------------------ */
/* Ontology scan + per-frame top-k (run_visual_tokenization.py:276,298-308).  */
/* img f32 [NF,D] ; txt f32 [NCpad,D] where each category c occupies rows     */
/* seg_start[c] .. seg_start[c]+seg_len[c]-1 and seg_start[c] % 32 == 0.      */
/* scores are exact f32: s = fma(a[k],b[k],s) for k = 0..D-1 in order.        */
/* out_index i32 [NF,ncat,topk] = class index WITHIN the category, ordered by */
/* (score desc, index asc); out_score f32 same shape.                         */
/* partial: caller workspace, vidil_scan_topk_ws_bytes() bytes.               */
/* ------------------------------------------------------------------------ */
int64_t vidil_scan_topk_ws_bytes(int32_t NF, int32_t NCpad, int32_t topk);
int vidil_scan_topk(const float* img, const float* txt, int32_t NF, int32_t D,
                    int32_t ncat, const int32_t* seg_start_host,
                    const int32_t* seg_len_host, int32_t topk, void* partial,
                    int32_t* out_index, float* out_score, void* stream);

/* ------------------------------------------------------------------------ */
/* BLIP retrieval backend of the visual tokenizer (--encoder_version blip,    */
/* run_visual_tokenization.py:277-293): `sims_matrix = image_embeds @          */
/* text_embeds.t()` then `sims.topk(k_test)` per frame before the ITM re-rank. */
/*   vidil_scan_scores: out f32 [NF,NC] = img [NF,D] · txt [NC,D]^T with the    */
/*     same exact k-ordered f32 chain as vidil_scan_topk;                      */
/*   vidil_topk_rows: the k (<= 128) largest of each of R rows of N (<= 38400)  */
/*     values, sorted by (value desc, index asc): out_v f32 [R,k], out_i i32.  */
/* ------------------------------------------------------------------------ */
int vidil_scan_scores(const float* img, const float* txt, int32_t NF, int32_t D,
                      int32_t NC, float* out, void* stream);
int vidil_topk_rows(const float* x, int64_t row_stride, int32_t R, int32_t N,
                    int32_t k, float* out_v, int32_t* out_i, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDIL_HIP_H */
