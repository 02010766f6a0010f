"""Weight packing helpers: fp32 ``nn.Parameter``s (reference checkpoint names) ->
device-resident f16 GEMM operands + f32 vectors the HIP kernels read.  Packing is
one-time per weight version (cached by a data_ptr/_version fingerprint)."""
from __future__ import annotations

import os

import torch

FP8 = torch.float8_e4m3fn      # OCP e4m3: the fp8 tower mode's GEMM operand type (gfx950's native fp8)
_DTYPES = {"f16": torch.float16, "fp16": torch.float16, "float16": torch.float16, "half": torch.float16,
           "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp8": FP8, "f8": FP8, "e4m3": FP8}


def as_compute_dtype(d):
    """'f16' / 'bf16' / 'fp8' / torch dtype -> torch.float16, torch.bfloat16 or torch.float8_e4m3fn.
    fp8 is a TOWER mode (BASELINE config 5): the four big GEMMs of every ViT / CLIP-vision block run on fp8 operands
    (2x the MFMA rate); attention, the text stacks and everything else stay in the 16-bit companion type
    ($VIDIL_FP8_COMPANION, f16 by default)."""
    if isinstance(d, str):
        if d.lower() not in _DTYPES:
            raise ValueError(f"unknown compute dtype {d!r} (f16, bf16 or fp8)")
        d = _DTYPES[d.lower()]
    if d not in (torch.float16, torch.bfloat16, FP8):
        raise ValueError(f"compute dtype must be float16, bfloat16 or float8_e4m3fn, got {d}")
    return d


def fp8_companion():
    d = as_compute_dtype(os.environ.get("VIDIL_FP8_COMPANION", "f16"))
    if d == FP8:
        raise ValueError("VIDIL_FP8_COMPANION must be f16 or bf16")
    return d


def w8(*weights):
    """nn.Linear weights (concatenated along N) -> (e4m3 [N,K] = W / scale[n], scale f32 [N]) with the largest weight
    of every output row at half of e4m3's range (224 of 448): the GEMM epilogue multiplies the accumulator by scale[n]."""
    w = (weights[0] if len(weights) == 1 else torch.cat(list(weights), dim=0)).detach().float()
    scale = (w.abs().amax(dim=1) / 224.0).clamp_min(1e-12)
    q = (w / scale[:, None]).to(FP8).contiguous()
    return q, scale.contiguous()


_default = [as_compute_dtype(os.environ.get("VIDIL_DTYPE", "f16"))]


def set_compute_dtype(dtype, *modules):
    """Select the 16-bit MFMA operand type — weights are re-packed and every activation buffer follows them.
    Without modules: the process-wide default ($VIDIL_DTYPE, f16 if unset) used by models that were not set
    individually.  With modules: those models (and all their sub-modules) only."""
    dtype = as_compute_dtype(dtype)
    if not modules:
        _default[0] = dtype
        return dtype
    for m in modules:
        for sub in m.modules():
            sub.__dict__["_compute_dtype"] = dtype
    return dtype


def compute_dtype(module=None):
    d = None if module is None else module.__dict__.get("_compute_dtype")
    return d if d is not None else _default[0]


# ---- precision mode -------------------------------------------------------------------------------------------------
# "parity": every GEMM on the caption path runs with ERROR-COMPENSATED operands — activations are handed over as
# [hi | lo | hi] rows (hi = T16(x), lo = T16(x - hi): vidil_split3_f32 / VIDIL_DT_SPLIT3 outputs), weights are packed as
# [W_hi | W_hi | W_lo], and ONE GEMM with K tripled accumulates x_hi·W_hi + x_lo·W_hi + x_hi·W_lo in f32, i.e. x·W to
# ~2^-21 relative instead of the 2^-11 of plain f16 operands.  It is the mode in which BASELINE's "caption logits within
# 1e-3" holds as an ABSOLUTE bound (tests/test_models_gpu.py) at ~3x the MFMA work; the throughput modes
# (plain f16 / bf16 / fp8) stay the default.  Attention in the mode (set_parity_attention): since round 5 the split-operand form —
# f32 Q / K / V from the projection GEMMs, every operand of Q.K^T and P.V as hi + lo on the 16-bit MFMA — by default; round 4's
# plain f32 arithmetic and round 3's 16-bit kernels (Q, K, V and the probabilities rounded to 16 bits) remain selectable.
_parity_default = [os.environ.get("VIDIL_PARITY", "0") == "1"]
_warned_bf16_parity = [False]
_PARITY_ATTN_KINDS = ("split", "f32", "16")
_parity_attn_default = [os.environ.get("VIDIL_PARITY_ATTN", "split") if os.environ.get("VIDIL_PARITY_ATTN", "split") in _PARITY_ATTN_KINDS
                        else "split"]


def set_parity_mode(on, *modules):
    """Switch the error-compensated "parity" precision mode on / off — process-wide without modules, else for those
    models (and all their sub-modules).  Weights are re-packed on the next call."""
    on = bool(on)
    if not modules:
        _parity_default[0] = on
        return on
    for m in modules:
        for sub in m.modules():
            sub.__dict__["_parity"] = on
    return on


def parity_mode(module=None) -> bool:
    v = None if module is None else module.__dict__.get("_parity")
    return _parity_default[0] if v is None else v


def set_parity_attention(kind, *modules):
    """Which attention the parity precision mode of ``modules`` (and their sub-modules) uses:
      "split" (the default since round 5) — vidil_attention_f32 with arith = 1: f32 Q / K / V read in place, every operand of the
        two contractions handed to the 16-bit MFMA as hi + lo (three products per contraction, f32 softmax) — the trick the
        mode's GEMMs use; the decode steps' cross-attention keeps the 16-bit K / V fragment tiles of the plain path (bytes are what
        bound it) and splits Q and the probabilities only.  Caption logits ~3e-5 of the LOGIT SCALE from the fp32 reference at
        trained-like statistics (tests/test_trained_like_gpu.py), at a fifth of the f32 kernels' cost;
      "f32" (round 4) — plain f32 arithmetic on f32 Q / K / V everywhere: ~1e-5 of the logit scale, ~1/16 of the MFMA kernels' rate;
      "16" (round 3) — the 16-bit MFMA attention kernels with [hi | lo | hi] outputs: 2.4e-4 of the logit scale, because Q / K / V
        are rounded to 16 bits inside them.
    Without modules: the process-wide default ($VIDIL_PARITY_ATTN)."""
    if kind is None:                      # drop the per-module choice again: the modules follow the process-wide default
        for m in modules:
            for sub in m.modules():
                sub.__dict__.pop("_parity_attn", None)
        return _parity_attn_default[0]
    if kind not in _PARITY_ATTN_KINDS:
        raise ValueError("parity attention: 'split', 'f32' or '16'")
    if not modules:
        _parity_attn_default[0] = kind
        return kind
    for m in modules:
        for sub in m.modules():
            sub.__dict__["_parity_attn"] = kind
    return kind


def parity_attention_kind(module=None) -> str:
    """"split", "f32" or "16" (see set_parity_attention)."""
    v = None if module is None else module.__dict__.get("_parity_attn")
    return _parity_attn_default[0] if v is None else v


def parity_attention_f32(module=None) -> bool:
    """True when the parity precision mode of ``module`` keeps Q / K / V as f32 rows and runs vidil_attention_f32 on them (kinds
    "split" and "f32"; see set_parity_attention)."""
    return parity_attention_kind(module) != "16"


def parity_attention_arith(module=None) -> int:
    """vidil_attention_f32's ``arith`` for this module: 1 (split-operand MFMA) or 0 (f32 arithmetic)."""
    return 1 if parity_attention_kind(module) == "split" else 0


def w3(*weights, dtype=None):
    """nn.Linear weights (concatenated along N) [N,K] f32 -> 16-bit [N,3K] = [W_hi | W_hi | W_lo] (hi = T16(W), lo =
    T16(W - hi)): the weight side of an error-compensated GEMM whose activation rows are [x_hi | x_lo | x_hi]."""
    w = (weights[0] if len(weights) == 1 else torch.cat(list(weights), dim=0)).detach().float()
    dtype = dtype or _default[0]
    hi = w.to(dtype)
    lo = (w - hi.float()).to(dtype)
    return torch.cat([hi, hi, lo], dim=1).contiguous()


def w3_patch(conv_weight, dtype=None):
    """w3 of a patch-embedding conv weight, K zero padded to a multiple of 64 like w16_patch (per plane)."""
    w = conv_weight.detach().reshape(conv_weight.shape[0], -1).float()
    K = w.shape[1]
    Kp = (K + 63) // 64 * 64
    if Kp != K:
        w = torch.nn.functional.pad(w, (0, Kp - K))
    return w3(w, dtype=dtype)


def fingerprint(module) -> tuple:
    return tuple((p.data_ptr(), p._version, p.device.index) for p in module.parameters()) + (compute_dtype(module), parity_mode(module))


def w16(*weights, dtype=None):
    """Concatenate nn.Linear weights along N and cast to the contiguous 16-bit GEMM operand [N,K]."""
    w = weights[0] if len(weights) == 1 else torch.cat(list(weights), dim=0)
    return w.detach().to(dtype or _default[0]).contiguous()


def w16_patch(conv_weight, dtype=None):
    """Patch-embedding conv weight [N,3,ps,ps] -> f16 [N, round_up(3*ps*ps, 64)], zero padded along K so the GEMM's
    K % 64 == 0 contract holds for any patch size (CLIP ViT-L/14: 588 -> 640; the patch rows are padded alike)."""
    w = conv_weight.detach().reshape(conv_weight.shape[0], -1).to(dtype or _default[0])
    K = w.shape[1]
    Kp = (K + 63) // 64 * 64
    if Kp != K:
        w = torch.nn.functional.pad(w, (0, Kp - K))
    return w.contiguous()


def fold_layernorm(weight, bias, gamma, beta, dtype=None):
    """LayerNorm(x; gamma, beta) followed by Linear(weight, bias), folded for the LN-fused GEMM
    (vidil_gemm_args.ln_fold): returns (W' = T16(gamma (.) W) [N,K], b' = b + W·beta f32 [N], colsum[n] = sum_k W'[n][k]
    f32 [N]).  colsum is taken from the ROUNDED W' — the kernel computes rstd * (x16·W'^T - mean * colsum) + b', which
    centres exactly the values the MFMA multiplies."""
    w = weight.detach().float()
    wf = (w * gamma.detach().float()[None, :]).to(dtype or _default[0]).contiguous()
    b = w @ beta.detach().float()
    if bias is not None:
        b = b + bias.detach().float()
    colsum = wf.double().sum(dim=1).float()
    return wf, b.contiguous(), colsum.contiguous()


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
    """Mixin: ``self.packed()`` returns the cached result of ``self._pack()`` (re-packed when a parameter or the
    compute dtype changes); ``self.cdt`` is the model's 16-bit operand type."""

    @property
    def cdt(self):
        """The 16-bit operand type of this model (in the fp8 tower mode: the companion type of everything that is not
        one of the towers' big GEMMs)."""
        d = compute_dtype(self)
        return fp8_companion() if d == FP8 else d

    @property
    def fp8(self):
        return compute_dtype(self) == FP8

    @property
    def parity(self):
        """Error-compensated GEMM operands everywhere on this model's path (set_parity_mode / $VIDIL_PARITY)."""
        on = parity_mode(self)
        if on and self.fp8:
            raise ValueError("the parity precision mode needs a 16-bit compute dtype (f16 is its intended type), not fp8")
        if on and compute_dtype(self) == torch.bfloat16 and not _warned_bf16_parity[0]:
            #  hi + lo of two bf16 values carries 16 significant bits, not f16's 22: the compensated product is good
            # to ~2^-16 relative and "caption logits within 1e-3" (an f16 statement) is NOT asserted for it
            import warnings

            warnings.warn("parity precision mode with bf16 operands: hi + lo carries 16 significant bits (f16: 22) — the "
                          "documented absolute 1e-3 logit bound holds for f16 operands only; use set_compute_dtype('f16', model)")
            _warned_bf16_parity[0] = True
        return on

    def pack_flags(self):
        """Host-side switches that change what ``_pack`` produces (overridden by the models that have any)."""
        return ()

    def packed(self):
        fp = fingerprint(self) + tuple(self.pack_flags())
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
