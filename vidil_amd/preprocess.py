"""Frame preprocessing on the GPU: the resize the reference does with PIL on the host, batched over frames.

  * ``blip_frames``  — ``transforms.Resize((S, S), interpolation=BICUBIC)`` of ``process_frame``
    (run_video_CapFilt.py:128-134): squash to S x S;
  * ``clip_frames``  — HF ``CLIPProcessor`` image side (run_visual_tokenization.py:138-142): shortest edge -> S with
    bicubic, centre crop S x S.
Both are ``PIL.Image.resize(size, BICUBIC)``, i.e. Pillow's two-pass antialiased resampling in 22-bit fixed point
(Resample.c).  The filter-weight set-up below follows ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` in double
precision on the host (a few thousand weights per frame geometry, cached); the passes themselves run in
``vidil_resample_u8`` and are bit-exact with Pillow.  ``/255`` and mean/std normalisation stay fused into the
patchify kernel that follows (``VisionTransformer.forward_u8`` / ``CLIPModel.encode_image_u8``).

Only the output columns / rows that survive the centre crop are computed.
"""
from __future__ import annotations

import math
from functools import lru_cache

import torch

from . import kernels as K
from .packing import require_cuda

_PRECISION_BITS = 32 - 8 - 2   # Pillow: 8-bit data, 2 bits of head room for the overshoot of the cubic


def _cubic(x: float) -> float:
    """Keys cubic, a = -0.5 (Pillow's BICUBIC)."""
    x = abs(x)
    if x < 1.0:
        return (1.5 * x - 2.5) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * -0.5
    return 0.0


@lru_cache(maxsize=256)
def axis_weights(in_size: int, out_size: int, first: int = 0, count: int | None = None):
    """Fixed-point weights of output indices first..first+count-1 along one axis.

    Returns (ksize, bounds, coeffs) as nested python lists: bounds[i] = (first source index, taps),
    coeffs[i] = ksize ints (zero padded)."""
    count = out_size - first if count is None else count
    scale = in_size / out_size
    fscale = max(scale, 1.0)                      # antialias: widen the kernel when shrinking
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / fscale
    one = 1 << _PRECISION_BITS
    bounds, coeffs = [], []
    for o in range(first, first + count):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        w = [_cubic((x - center + 0.5) * inv) for x in range(lo, hi)]
        total = 0.0
        for v in w:
            total += v
        if total != 0.0:
            w = [v / total for v in w]
        fixed = [int(v * one - 0.5) if v < 0 else int(v * one + 0.5) for v in w]
        bounds.append((lo, hi - lo))
        coeffs.append(fixed + [0] * (ksize - len(fixed)))
    return ksize, bounds, coeffs


class ResizePlan:
    """Device-resident weight tables for one (input size -> output size [, crop]) geometry."""

    def __init__(self, in_h, in_w, out_h, out_w, crop, device):
        top, left, ch, cw = crop if crop is not None else (0, 0, out_h, out_w)
        if not (0 <= top and top + ch <= out_h and 0 <= left and left + cw <= out_w):
            raise ValueError(f"crop {crop} outside the {out_h}x{out_w} resized frame")
        self.in_h, self.in_w, self.ch, self.cw = in_h, in_w, ch, cw
        self.need_h = out_w != in_w
        self.need_v = out_h != in_h
        i32 = dict(dtype=torch.int32, device=device)
        _, bv, kv = axis_weights(in_h, out_h, top, ch)
        if self.need_v:
            self.row0 = bv[0][0]
            self.rows = bv[-1][0] + bv[-1][1] - self.row0          # source rows the vertical pass reads
            self.bv = torch.tensor([(lo - self.row0, n) for lo, n in bv], **i32)
            self.kv = torch.tensor(kv, **i32)
        else:                                                       # same height: the crop is a row slice
            self.row0, self.rows = top, ch
        if self.need_h:
            _, bh, kh = axis_weights(in_w, out_w, left, cw)
            self.bh = torch.tensor(bh, **i32)
            self.kh = torch.tensor(kh, **i32)
        else:
            self.col0 = left

    def run(self, frames_u8):
        N = frames_u8.shape[0]
        dev = frames_u8.device
        cur = frames_u8
        if self.need_h:
            tmp = torch.empty((N, self.rows, self.cw, 3), dtype=torch.uint8, device=dev)
            K.resample_u8(cur, tmp, self.bh, self.kh, vertical=False, src_row0=self.row0)
            cur = tmp
        else:
            cur = cur[:, self.row0:self.row0 + self.rows, self.col0:self.col0 + self.cw].contiguous()
        if self.need_v:
            out = torch.empty((N, self.ch, self.cw, 3), dtype=torch.uint8, device=dev)
            K.resample_u8(cur, out, self.bv, self.kv, vertical=True)
            cur = out
        return cur


_plans = {}


def _plan(in_h, in_w, out_h, out_w, crop, device):
    key = (in_h, in_w, out_h, out_w, crop, str(device))
    p = _plans.get(key)
    if p is None:
        p = _plans[key] = ResizePlan(in_h, in_w, out_h, out_w, crop, device)
    return p


def _check(frames_u8, who):
    require_cuda(frames_u8, who)
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
        raise K.VidilHipError(f"{who}: expected uint8 [N,H,W,3] frames, got {frames_u8.dtype} {tuple(frames_u8.shape)}")
    return frames_u8.contiguous()


def blip_frames(frames_u8, S):
    """uint8 [N,H,W,3] on the GPU -> uint8 [N,S,S,3]  (run_video_CapFilt.py:128-134 up to ToTensor)."""
    f = _check(frames_u8, "blip_frames")
    _, H, W, _ = f.shape
    if (H, W) == (S, S):
        return f
    return _plan(H, W, S, S, None, f.device).run(f)


def clip_resized_hw(H, W, S):
    """HF CLIP feature extractor: shortest edge -> S, the other edge int(S * long / short)."""
    if W <= H:
        return int(S * H / W), S
    return S, int(S * W / H)


def clip_frames(frames_u8, S=224):
    """uint8 [N,H,W,3] on the GPU -> uint8 [N,S,S,3]  (HF CLIPProcessor: resize shortest edge, centre crop)."""
    f = _check(frames_u8, "clip_frames")
    _, H, W, _ = f.shape
    if (H, W) == (S, S):
        return f
    nh, nw = clip_resized_hw(H, W, S)
    return _plan(H, W, nh, nw, ((nh - S) // 2, (nw - S) // 2, S, S), f.device).run(f)
