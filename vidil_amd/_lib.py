"""ctypes binding of libvidil_hip.so (the C ABI declared in include/vidil_hip.h).

The product path has no CPU fallback: if the shared library is missing this
module raises at first use, loudly.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C vidil_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede the CDLL below: torch ships its own libamdhip64; loading ours
# first would bind libvidil_hip.so to a second HIP runtime that has no initialised device.

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIDIL_HIP_LIB: developer override (A/B-ing two builds of the same ABI on one box)
LIB_PATH = os.environ.get("VIDIL_HIP_LIB") or os.path.join(_HERE, "csrc", "libvidil_hip.so")

EPI_F16, EPI_F32, EPI_HEADS, EPI_PATCH, EPI_ARENA, EPI_F8 = 0, 1, 2, 3, 4, 5
DT_F16, DT_BF16, DT_FP8 = 0, 1, 2
DT_SPLIT3 = 0x100     # output flag: rows written as error-compensated operands [hi | lo | hi] (include/vidil_hip.h)
DT_SPLIT2 = 0x200      # with DT_SPLIT3: planes hi | lo written only (include/vidil_hip.h VIDIL_DT_SPLIT2)
ACT_NONE, ACT_GELU_ERF, ACT_QUICK_GELU = 0, 1, 2


class GemmArgs(C.Structure):
    """Mirror of ``vidil_gemm_args`` (include/vidil_hip.h)."""

    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32),
        ("epi", C.c_int32), ("act", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("resid", C.c_void_p),
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p),
        ("T", C.c_int32), ("H", C.c_int32), ("part0", C.c_int32), ("t_off", C.c_int32),
        ("Tq_cap", C.c_int32), ("Tk_cap", C.c_int32), ("NP", C.c_int32),
        ("q_scale", C.c_float),
        ("arena_rows", C.c_int32), ("slot_stride", C.c_int32),
        ("pos", C.c_void_p), ("tpi", C.c_int32), ("kv_tiled", C.c_int32), ("dtype", C.c_int32),
        ("out16", C.c_void_p), ("ldo16", C.c_int32), ("ln_fold", C.c_int32), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("ln_stats_out", C.c_void_p), ("ln_stats", C.c_void_p),
        ("w_scale", C.c_void_p), ("dtype16", C.c_int32), ("rln_gamma", C.c_void_p), ("rln_beta", C.c_void_p),
        ("out16_split3", C.c_int32), ("split_k", C.c_int32),
    ]


class BeamState(C.Structure):
    """Mirror of ``vidil_beam_state``."""

    _fields_ = [(n, C.c_void_p) for n in (
        "seqs", "seqs_next", "beam_scores", "beam_idx", "next_tok", "done", "n_hyp",
        "hyp_score", "hyp_len", "hyp_tok", "worst", "n_done")]


class AttnF32Args(C.Structure):
    """Mirror of ``vidil_attn_f32_args`` (include/vidil_hip.h; field order and types are the ABI)."""

    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
        ("q_off", C.c_int32), ("k_off", C.c_int32), ("v_off", C.c_int32),
        ("out_mode", C.c_int32), ("dtype16", C.c_int32),
        ("Bq", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("kv_rows", C.c_int32),
        ("kv_group", C.c_int32),
        ("kv_index", C.c_void_p), ("group_start", C.c_void_p),
        ("n_kv", C.c_int32), ("max_group", C.c_int32),
        ("kv_len", C.c_void_p),
        ("causal", C.c_int32), ("causal_off", C.c_int32),
        ("anc", C.c_void_p),
        ("anc_ld", C.c_int32), ("arena_rows", C.c_int32),
        ("scale", C.c_float),
        ("arith", C.c_int32), ("kv16", C.c_int32),
    ]


_i32, _i64, _f32, _p = C.c_int32, C.c_int64, C.c_float, C.c_void_p

# name -> (restype, argtypes); the order/types restate include/vidil_hip.h.
SIGNATURES = {
    "vidil_last_error": (C.c_char_p, []),
    "vidil_abi_version": (_i32, []),
    "vidil_num_entry_points": (_i32, []),
    "vidil_gemm": (_i32, [C.POINTER(GemmArgs), _p]),
    "vidil_gemm_kernel_name": (_i32, [C.POINTER(GemmArgs), C.c_char_p, _i32]),
    "vidil_gemm_split_k_in_loop": (_i32, []),
    "vidil_gemm_split_k_serves": (_i32, [C.POINTER(GemmArgs)]),
    "vidil_layernorm": (_i32, [_p, _i64, _p, _p, _f32, _i32, _i32, _p, _i32, _p, _p]),
    "vidil_split3_f32": (_i32, [_p, _p, _i32, _i32, _i32, _p]),
    "vidil_attention": (_i32, [_p, _p, _p, _p, _p, _p, _p] + [_i32] * 16 + [_p]),
    "vidil_attention_f32": (_i32, [C.POINTER(AttnF32Args), _p]),
    "vidil_patchify_f32": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p]),
    "vidil_patchify_u8": (_i32, [_p, _p, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _i32, _p]),
    "vidil_resample_u8": (_i32, [_p, _p] + [_i32] * 6 + [_p, _p, _i32, _i32, _p]),
    "vidil_set_cls_row": (_i32, [_p, _p, _p, _i32, _i32, _i32, _p]),
    "vidil_embed_tokens": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "vidil_gather_rows_f32": (_i32, [_p, _p, _p, _i32, _i32, _p]),
    "vidil_l2_normalize_rows": (_i32, [_p, _i32, _i32, _p]),
    "vidil_logsoftmax_topk": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "vidil_logsoftmax_topk_penalty": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p, _i32, _i32, C.c_float, _p, _p, _p]),
    "vidil_beam_update": (_i32, [C.POINTER(BeamState), _p, _p] + [_i32] * 7 + [_p]),
    "vidil_beam_finalize": (_i32, [C.POINTER(BeamState)] + [_i32] * 6 + [_p, _p, _p, _p]),
    "vidil_beam_ancestry": (_i32, [_p, _p, _p, _i32, _i32, _i32, _p]),
    "vidil_beam_attention": (_i32, [_p, _p, _p, _p, _p] + [_i32] * 8 + [_p]),
    "vidil_sample_top_k_top_p": (_i32, [_p, _p, _p, _p, _p] + [_i32] * 8 + [_f32, _f32, C.c_uint64, _i32, _i32, _p]),
    "vidil_scan_scores": (_i32, [_p, _p, _i32, _i32, _i32, _p, _p]),
    "vidil_topk_rows": (_i32, [_p, _i64, _i32, _i32, _i32, _p, _p, _p]),
    "vidil_scan_topk_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "vidil_scan_topk": (_i32, [_p, _p, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), _i32, _p, _p, _p, _p]),
}

_lib = None


class VidilHipError(RuntimeError):
    pass


ABI_VERSION = 12     # include/vidil_hip.h as this binding mirrors it (struct layouts, argument lists)


def load():
    """Load (once) and return the ctypes library with typed entry points; a library of another ABI version is refused."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VidilHipError(
            f"{LIB_PATH} is missing: the HIP extension is not built. There is no CPU "
            "fallback for the product path. Run `make -C vidil_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.vidil_abi_version() != ABI_VERSION:
        raise VidilHipError(f"{LIB_PATH} implements ABI {lib.vidil_abi_version()}, this binding ABI {ABI_VERSION}: rebuild it "
                            "(`make -C vidil_amd/csrc`)")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    """Map a non-zero return code to an exception carrying vidil_last_error()."""
    if rc != 0:
        msg = load().vidil_last_error().decode("utf-8", "replace")
        raise VidilHipError(f"{what or 'vidil'} failed (rc={rc}): {msg}")
