"""oracle/scan_ref.c against the reference FORM of the ontology scan (run_visual_tokenization.py:276,298-308):
`image_embeds @ text_embeds.t()` and `np.argsort(score)[::-1][:5]`, on the full vg-sized layout (42,759 classes)."""
import ctypes
import os

import numpy as np

import scan_cases as sc
from common import ROOT


def _lib():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libscan_ref.so"))
    assert hasattr(lib, "vidil_ref_scores")
    return lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def test_scan_ref_scores_and_topk_equal_the_numpy_reference_form_on_the_vg_layout():
    lib = _lib()
    NF, D, TOPK = 24, 512, 5
    emb, texts = sc.vg_layout(D)
    img = sc.frames(NF, D, emb=emb, near=8)
    ref_texts, s32, s64 = sc.reference_form(img, emb, texts, TOPK)
    mat, seg_start, seg_len = sc.packed(emb)
    # (1) scores: the oracle's k-ordered fmaf chain vs numpy's f32 matmul vs f64
    max_err_oracle = max_err_np = 0.0
    for k, s0, n in zip(sc.CATS, seg_start, seg_len):
        out = np.zeros((NF, n), np.float32)
        lib.vidil_ref_scores(_p(img), _p(mat[s0:s0 + n]), NF, D, n, _p(out))
        max_err_oracle = max(max_err_oracle, float(np.abs(out - s64[k]).max()))
        max_err_np = max(max_err_np, float(np.abs(s32[k] - s64[k]).max()))
    # unit-norm operands, 512 terms: the a-priori bound is 512 * 2^-24 = 3e-5; measured errors are 3-5e-7
    assert max_err_oracle < 3e-6 and max_err_np < 3e-6, (max_err_oracle, max_err_np)
    tau = 2.0 * (max_err_oracle + max_err_np)
    # (2) top-5 as texts: identical to the reference form wherever the reference's own ranking is decided at tau
    ri = np.zeros((NF, 4, TOPK), np.int32)
    rs = np.zeros((NF, 4, TOPK), np.float32)
    lib.vidil_ref_scan_topk(_p(img), _p(mat), NF, D, 4, (ctypes.c_int32 * 4)(*seg_start), (ctypes.c_int32 * 4)(*seg_len), TOPK,
                            _p(ri), _p(rs))
    masked = mismatched_outside = 0
    for f in range(NF):
        for c, k in enumerate(sc.CATS):
            got = [texts[k][int(i)] for i in ri[f, c]]
            if sc.undecided(s64[k][f], tau, TOPK):
                masked += 1
                continue
            mismatched_outside += int(got != ref_texts[k][f])
    print(f"scan_ref vs numpy: max|err| oracle {max_err_oracle:.2e}, numpy f32 {max_err_np:.2e}, tau {tau:.2e}, "
          f"undecided rows {masked}/{NF * 4}")
    assert mismatched_outside == 0
    assert masked <= NF * 4 // 20          # the mask is a handful of near-ties, not an escape hatch


def test_tie_rule_lower_index_first_vs_argsort_on_duplicates_emits_the_same_texts():
    lib = _lib()
    D = 64
    rng = np.random.default_rng(3)
    e = rng.standard_normal((40, D)).astype(np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    e[7] = e[3]; e[21] = e[3]                                  # three copies of class 3's row
    texts = [f"t{i}" for i in range(40)]
    texts[7] = texts[21] = texts[3]
    img = (e[3] * 2 + rng.standard_normal(D).astype(np.float32) * 0.01)[None]
    img = (img / np.linalg.norm(img)).astype(np.float32)
    mat = np.zeros((64, D), np.float32); mat[:40] = e
    ri = np.zeros((1, 1, 5), np.int32); rs = np.zeros((1, 1, 5), np.float32)
    lib.vidil_ref_scan_topk(_p(img), _p(mat), 1, D, 1, (ctypes.c_int32 * 1)(0), (ctypes.c_int32 * 1)(40), 5, _p(ri), _p(rs))
    assert ri[0, 0, :3].tolist() == [3, 7, 21]                 # ties -> lower index first
    ref = [texts[i] for i in np.argsort((img @ e.T)[0])[::-1][:5]]
    assert [texts[i] for i in ri[0, 0]] == ref
