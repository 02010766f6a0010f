"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the frame resize the reference applies
before its models, i.e. Pillow's antialiased bicubic ``Image.resize`` on 8-bit RGB.

Where the reference calls it
  * run_video_CapFilt.py:128-134 — ``transforms.Resize((S, S), interpolation=BICUBIC)`` on a PIL image, which is
    ``img.resize((S, S), Image.BICUBIC)`` (torchvision functional_pil.resize), then ToTensor + Normalize;
  * run_visual_tokenization.py:138-142 — HF ``CLIPProcessor``: resize the SHORTEST edge to 224 with
    ``Image.BICUBIC`` (long edge = int(224 * long / short)), centre-crop 224, 1/255, normalise.
Both are third-party (Pillow / torchvision / transformers, unpinned in docker/requirements.txt) and absent from
/root/reference, so the algorithm is restated from Pillow's published ``src/libImaging/Resample.c``
(``precompute_coeffs``, ``normalize_coeffs_8bpc``, ``ImagingResampleHorizontal_8bpc`` / ``Vertical_8bpc``,
``ImagingResample``) — stable since Pillow 7 — and PINNED bit for bit against the Pillow installed in this image
(tests/test_resize_cpu.py, plus committed golden vectors made by tests/golden/make_resize_golden.py).

Everything after the double-precision coefficient set-up is integer arithmetic, so parity is BIT-EXACT:
  k_fixed = (int)(k * 2^22 -+ 0.5)           (truncation toward zero, like the C cast)
  out     = clip8((2^21 + sum_i pixel_i * k_fixed_i) >> 22)
horizontal pass first (only over the source rows the vertical pass will read), u8 intermediate, then vertical.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BICUBIC_SUPPORT = 2.0


def bicubic_filter(x: float) -> float:
    """Resample.c bicubic_filter, a = -0.5 (Keys)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, in0: float, in1: float, out_size: int):
    """Resample.c precompute_coeffs: (ksize, bounds int [out,2] = (first tap, tap count), kk float64 [out,ksize])."""
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ww = 0.0
        xmin = int(center - support + 0.5)      # C (int) cast: toward zero (the argument is > -1 here)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        for x in range(xmax):
            w = bicubic_filter((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def normalize_coeffs_8bpc(kk: np.ndarray) -> np.ndarray:
    """Resample.c normalize_coeffs_8bpc: fixed point with 22 fractional bits, C cast = truncation."""
    out = np.empty(kk.shape, dtype=np.int64)
    flat_in, flat_out = kk.reshape(-1), out.reshape(-1)
    for i, v in enumerate(flat_in):
        flat_out[i] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
    return out


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _pass(src: np.ndarray, bounds: np.ndarray, kfix: np.ndarray, axis: int) -> np.ndarray:
    """One separable pass along ``axis`` (0 = rows / vertical, 1 = columns / horizontal) of src u8 [H,W,C]."""
    n_out = bounds.shape[0]
    shape = list(src.shape)
    shape[axis] = n_out
    out = np.empty(shape, dtype=np.uint8)
    s = src.astype(np.int64)
    for o in range(n_out):
        lo, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        k = kfix[o, :cnt]
        if axis == 1:
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(s[:, lo:lo + cnt, :], k, axes=([1], [0]))
            out[:, o, :] = _clip8(acc)
        else:
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(s[lo:lo + cnt, :, :], k, axes=([0], [0]))
            out[o, :, :] = _clip8(acc)
    return out


def resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``PIL.Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)`` for u8 [H,W,3] (ImagingResample)."""
    in_h, in_w = img.shape[:2]
    if (in_w, in_h) == (out_w, out_h):
        return img.copy()
    need_h = out_w != in_w
    need_v = out_h != in_h
    _, bh, kh = precompute_coeffs(in_w, 0.0, float(in_w), out_w)
    _, bv, kv = precompute_coeffs(in_h, 0.0, float(in_h), out_h)
    cur = img
    bv = bv.copy()
    if need_h:
        first = int(bv[0, 0])
        last = int(bv[-1, 0] + bv[-1, 1])
        bv[:, 0] -= first
        cur = _pass(img[first:last], bh, normalize_coeffs_8bpc(kh), axis=1)
    if need_v:
        cur = _pass(cur, bv, normalize_coeffs_8bpc(kv), axis=0)
    return cur


def blip_process_frame_u8(img: np.ndarray, S: int) -> np.ndarray:
    """run_video_CapFilt.py:128-134 up to (not including) ToTensor: squash to S x S."""
    return resize_bicubic_u8(img, S, S)


def clip_output_size(in_h: int, in_w: int, S: int):
    """HF CLIP feature extractor: shortest edge -> S, the other = int(S * long / short)."""
    if in_w <= in_h:
        return int(S * in_h / in_w), S      # (new_h, new_w)
    return S, int(S * in_w / in_h)


def clip_process_frame_u8(img: np.ndarray, S: int = 224) -> np.ndarray:
    """HF CLIPProcessor image side up to the rescale: shortest-edge bicubic resize, centre crop S x S."""
    in_h, in_w = img.shape[:2]
    new_h, new_w = clip_output_size(in_h, in_w, S)
    r = resize_bicubic_u8(img, new_w, new_h)
    top, left = (new_h - S) // 2, (new_w - S) // 2
    return r[top:top + S, left:left + S].copy()
