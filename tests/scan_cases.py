"""Shared by the CPU and GPU scan-parity tests: the vg-sized synthetic ontology, the reference form of the scan
(run_visual_tokenization.py:276,298-308) in numpy, and the decidability mask."""
import numpy as np

VG_SIZES = dict(objects=19958, attributes=15026, scenes=365, verbs=7410)     # SURVEY.md §8 a26 (replayed on the JSONs)
CATS = ("objects", "attributes", "scenes", "verbs")


def vg_layout(dim=512, seed=0, sizes=VG_SIZES):
    """Unit-norm f32 class embeddings per category + class texts; 'scenes' carries duplicate strings like the real
    place365 list (25 x the same text -> identical rows -> exact score ties)."""
    rng = np.random.default_rng(seed)
    emb, texts = {}, {}
    for k in CATS:
        n = sizes[k]
        e = rng.standard_normal((n, dim)).astype(np.float32)
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        t = [f"{k}_{i}" for i in range(n)]
        if k == "scenes" and n > 130:
            for j in range(1, 25):
                e[40 + j] = e[40]; t[40 + j] = t[40]
                e[100 + j] = e[100]; t[100 + j] = t[100]
        emb[k], texts[k] = e.astype(np.float32), t
    return emb, texts


def frames(nf, dim=512, seed=1, emb=None, near=0):
    """Unit-norm f32 image embeddings.  ``near`` of them are placed very close to a class embedding so that the top
    scores are well separated in some rows and nearly tied in others."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((nf, dim)).astype(np.float32)
    if emb is not None:
        for i in range(min(near, nf)):
            k = CATS[i % 4]
            x[i] = emb[k][(i * 37) % emb[k].shape[0]] * 3.0 + x[i] * 0.05
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def packed(emb):
    """The scan's input layout: one f32 matrix, categories at multiples of 32 rows (zero rows between)."""
    seg_start, seg_len, n = [], [], 0
    for k in CATS:
        seg_start.append(n)
        seg_len.append(emb[k].shape[0])
        n += (emb[k].shape[0] + 31) // 32 * 32
    m = np.zeros((n, emb[CATS[0]].shape[1]), np.float32)
    for k, s in zip(CATS, seg_start):
        m[s:s + emb[k].shape[0]] = emb[k]
    return m, seg_start, seg_len


def reference_form(img, emb, texts, topk=5):
    """The reference, literally: scores = image_embeds @ text_embeds.t() (f32, whatever order the BLAS uses), then
    per frame np.argsort(score)[::-1][:topk] -> texts.  Also returns f64 scores for the decidability mask."""
    out_texts, s32, s64 = {}, {}, {}
    for k in CATS:
        s32[k] = img @ emb[k].T
        s64[k] = img.astype(np.float64) @ emb[k].astype(np.float64).T
        idx = np.argsort(s32[k], axis=1)[:, ::-1][:, :topk]
        out_texts[k] = [[texts[k][int(i)] for i in row] for row in idx]
    return out_texts, s32, s64


def undecided(s64_row, tau, topk=5):
    """True when the reference's own ranking of the top (topk+1) DISTINCT scores is not decided at resolution tau: some
    adjacent gap is positive but below tau (zero gaps are duplicate class rows, which emit the same text either way)."""
    top = np.sort(s64_row)[::-1][:topk + 26]      # enough to see past a run of 25 duplicates
    gaps = top[:-1] - top[1:]
    distinct = np.concatenate([[True], gaps > 0])
    vals = top[distinct][:topk + 1]
    g = vals[:-1] - vals[1:]
    return bool(np.any(g < tau))
