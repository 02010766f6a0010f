/*
 * oracle/scan_ref.c — TEST INFRASTRUCTURE, never linked into the product.
 *
 * CPU restatement (plain C) of the ontology scan + per-frame top-k of the
 * reference: run_visual_tokenization.py:276 (`image_embeds @ text_embeds.t()`)
 * and :301-308 (`np.argsort(frm_score)[::-1][:topk]`), per category.
 *
 * Two things are pinned here that the reference leaves to its BLAS / numpy
 * build: (1) the f32 summation order — the order the HIP kernel's f32 MFMA
 * chain uses, k = 8c + {0,4,1,5,2,6,3,7}, one fmaf per product — so scores are
 * bit-identical to the GPU's; (2) the tie rule — score descending, then class
 * index ascending (np.argsort's order among equal scores is unspecified; the
 * only exact ties on this path are duplicate class strings, which emit the
 * same text either way).
 *
 * What pins THIS file to the reference form: tests/test_scan_ref_cpu.py compares its
 * scores with numpy's `img @ txt.T` (f32 BLAS order and f64) and its top-k with
 * `np.argsort(row)[::-1][:k]` as class TEXTS on the full 42,759-class layout; they
 * are identical wherever the reference's own adjacent score gaps exceed the
 * summation-order error (the test reports how few rows are not decided).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -o oracle/_build/libscan_ref.so oracle/scan_ref.c -lm
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

float vidil_ref_score(const float* t, const float* f, int D) {
  float s = 0.0f;
  for (int c = 0; c < D / 8; ++c) {
    for (int j = 0; j < 4; ++j) {
      s = fmaf(t[8 * c + j], f[8 * c + j], s);
      s = fmaf(t[8 * c + 4 + j], f[8 * c + 4 + j], s);
    }
  }
  return s;
}

/* dense scores out[f][c] = score(txt[c], img[f]) — for the comparison with numpy's matmul */
void vidil_ref_scores(const float* img, const float* txt, int NF, int D, int NC, float* out) {
  for (int f = 0; f < NF; ++f)
    for (int c = 0; c < NC; ++c) out[(size_t)f * NC + c] = vidil_ref_score(txt + (size_t)c * D, img + (size_t)f * D, D);
}

static int better(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

/* img [NF,D], txt [NCpad,D]; category c = rows seg_start[c] .. +seg_len[c]-1.
 * out_index/out_score [NF,ncat,topk]; index is within the category, -1 if the
 * category has fewer than topk classes. */
void vidil_ref_scan_topk(const float* img, const float* txt, int NF, int D, int ncat, const int32_t* seg_start,
                         const int32_t* seg_len, int topk, int32_t* out_index, float* out_score) {
  for (int f = 0; f < NF; ++f) {
    for (int c = 0; c < ncat; ++c) {
      float bs[16];
      int bi[16];
      for (int j = 0; j < topk; ++j) { bs[j] = -INFINITY; bi[j] = 0x7fffffff; }
      for (int i = 0; i < seg_len[c]; ++i) {
        const float s = vidil_ref_score(txt + (size_t)(seg_start[c] + i) * D, img + (size_t)f * D, D);
        if (better(s, i, bs[topk - 1], bi[topk - 1])) {
          int j = topk - 1;
          while (j > 0 && better(s, i, bs[j - 1], bi[j - 1])) { bs[j] = bs[j - 1]; bi[j] = bi[j - 1]; --j; }
          bs[j] = s; bi[j] = i;
        }
      }
      for (int j = 0; j < topk; ++j) {
        out_index[((size_t)f * ncat + c) * topk + j] = bi[j] == 0x7fffffff ? -1 : bi[j];
        out_score[((size_t)f * ncat + c) * topk + j] = bs[j];
      }
    }
  }
}
