"""Weight packing helpers: fp32 ``nn.Parameter``s (reference checkpoint names) ->
device-resident f16 GEMM operands + f32 vectors the HIP kernels read.  Packing is
one-time per weight version (cached by a data_ptr/_version fingerprint)."""
from __future__ import annotations

import torch


def fingerprint(module) -> tuple:
    return tuple((p.data_ptr(), p._version, p.device.index) for p in module.parameters())


def w16(*weights):
    """Concatenate nn.Linear weights along N and cast to contiguous f16 [N,K]."""
    w = weights[0] if len(weights) == 1 else torch.cat(list(weights), dim=0)
    return w.detach().to(torch.float16).contiguous()


def w16_patch(conv_weight):
    """Patch-embedding conv weight [N,3,ps,ps] -> f16 [N, round_up(3*ps*ps, 64)], zero padded along K so the GEMM's
    K % 64 == 0 contract holds for any patch size (CLIP ViT-L/14: 588 -> 640; the patch rows are padded alike)."""
    w = conv_weight.detach().reshape(conv_weight.shape[0], -1).to(torch.float16)
    K = w.shape[1]
    Kp = (K + 63) // 64 * 64
    if Kp != K:
        w = torch.nn.functional.pad(w, (0, Kp - K))
    return w.contiguous()


def v32(*vecs):
    """Concatenate bias/LN vectors, contiguous f32 (None if every part is None)."""
    if all(v is None for v in vecs):
        return None
    parts = []
    for v in vecs:
        if v is None:
            raise ValueError("cannot concatenate a missing bias with present ones")
        parts.append(v.detach().to(torch.float32).reshape(-1))
    return (parts[0] if len(parts) == 1 else torch.cat(parts)).contiguous()


class PackedCache:
    """Mixin: ``self.packed()`` returns the cached result of ``self._pack()``."""

    def packed(self):
        fp = fingerprint(self)
        cache = self.__dict__.get("_packed_cache")
        if cache is None or cache[0] != fp:
            cache = (fp, self._pack())
            self.__dict__["_packed_cache"] = cache
        return cache[1]


def require_cuda(t, what):
    from ._lib import VidilHipError

    if not t.is_cuda:
        raise VidilHipError(
            f"{what}: input is on {t.device}; the vidil_amd product path runs only on an AMD GPU "
            "through libvidil_hip.so (no CPU fallback)")
