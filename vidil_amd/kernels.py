"""Thin tensor -> raw-pointer wrappers over the C ABI (include/vidil_hip.h).

torch is used here only for device memory and the current HIP stream; every
arithmetic op on the hot path is one of the HIP kernels behind these calls.
All functions raise ``VidilHipError`` on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import (ACT_GELU_ERF, ACT_NONE, ACT_QUICK_GELU, DT_BF16, DT_F16, DT_FP8, DT_SPLIT2, DT_SPLIT3, EPI_ARENA, EPI_F8, EPI_F16, EPI_F32,
                   EPI_HEADS, EPI_PATCH, BeamState, GemmArgs, VidilHipError, check)

__all__ = [
    "gemm", "layernorm", "attention", "patchify_f32", "patchify_u8", "set_cls_row",
    "embed_tokens", "gather_rows", "l2_normalize_rows", "logsoftmax_topk", "BeamBuffers",
    "beam_update", "beam_finalize", "scan_topk", "scan_topk_ws_bytes",
    "ACT_NONE", "ACT_GELU_ERF", "ACT_QUICK_GELU", "VidilHipError",
]


def vt_pos(t: int) -> int:
    """Storage column of key ``t`` in a V^T buffer (include/vidil_hip.h: 16-key blocks hold keys in the order
    0-3, 8-11, 4-7, 12-15).  An involution: it also maps a column back to its key."""
    return t ^ (12 if ((t >> 2) ^ (t >> 3)) & 1 else 0)


def vt_columns(n: int):
    """LongTensor c with c[t] = vt_pos(t) for t < n (for building / reading V^T buffers in tests)."""
    return torch.tensor([vt_pos(t) for t in range(n)], dtype=torch.long)


def kv_tile_offsets(n: int):
    """(k_off, v_off) LongTensors [n,64]: element offset of (key t, dim c) inside a (batch, head)'s fragment-tiled K / V
    buffer (csrc/common.h ktile_off / vtile_off) — for building / reading tiled buffers in tests."""
    t = torch.arange(n).view(n, 1)
    c = torch.arange(64).view(1, 64)
    k_off = (t >> 5) * 2048 + ((c >> 3) * 32 + (t & 31)) * 8 + (c & 7)
    tt, g = t & 31, (t & 15) >> 2
    v_off = (t >> 5) * 2048 + ((((tt >> 4) * 2 + (c >> 5)) * 2 + (g & 1)) * 32 + (c & 31)) * 8 + (g >> 1) * 4 + (tt & 3)
    return k_off, v_off


_DT = {torch.float16: DT_F16, torch.bfloat16: DT_BF16, torch.float8_e4m3fn: DT_FP8}


def _dt(t, name="tensor", fp8_ok=False) -> int:
    """VIDIL_DT_* code of an operand tensor (float16 / bfloat16; float8_e4m3fn where the entry point takes it)."""
    d = t if isinstance(t, torch.dtype) else t.dtype
    code = _DT.get(d)
    if code is None or (code == DT_FP8 and not fp8_ok):
        raise VidilHipError(f"{name}: expected a float16 or bfloat16{' or float8_e4m3fn' if fp8_ok else ''} tensor, got {d}")
    return code


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t, dtype=None, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise VidilHipError(f"{name}: expected a CUDA/HIP tensor, got {t.device} (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise VidilHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise VidilHipError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


# --------------------------------------------------------------------------- GEMM
def gemm(a, w, bias=None, **kw):
    """C = A · W^T with a fused epilogue.

    a [M,K], w [N,K] in the same 16-bit type T16 (float16 or bfloat16), bias f32 [N] or None.
      * default: returns/fills ``out`` [M,N] (T16, or f32 with ``out_dtype=torch.float32``); f32 may add ``resid``.
      * heads=dict(q=,k=,vt=,T=,H=,part0=,t_off=,Tq_cap=,Tk_cap=,NP=,q_scale=,tiled=): per-head scatter.
      * patch=dict(out=,pos=,tpi=): patch-embedding epilogue (row remap + pos embed).
      * arena=dict(q=,k=,v=,T=,H=,part0=,t_off=,arena_rows=,slot_stride=,Tcap=,q_scale=): Q rows + K/V rows
        appended to a beam-search KV arena [position][slot][H*64] (see vidil_beam_attention).
      * split3_out=T16 [M,3N]: f32 epilogue (activation in f32) whose result is written as the error-compensated operand rows
        [hi | lo | hi] of the next K-tripled GEMM (vidil_gemm_args.out16_split3) — the parity mode's fc1 -> fc2 hand-over
        without the f32 round trip through ``split3``; ``out`` (f32), if given as well, also receives the f32 rows.
      * split_k=True: ``a`` holds [x_hi | x_lo | x_hi] rows and ``w`` [W_hi | W_hi | W_lo] (packing.w3): the error-compensated
        product of the parity precision mode; the library may form the three products inside one K loop (vidil_gemm_args.split_k).
        ``a_planes=2``: only planes hi | lo of ``a`` were written (a ``planes=2`` / ``split3_planes=2`` producer) — the call fails
        with EINVAL instead of silently reading the third plane when it does not qualify for the K-loop form.
      * LayerNorm folded into a pre-LN block's GEMM pair (vidil_gemm_args.ln_fold): the residual GEMM passes
        ``out16=`` (T16 copy of the f32 stream it writes) and ``ln_stats_out=`` (per-row partial sums), the next
        GEMM passes that copy as ``a`` with ``ln=(colsum, eps, stats)`` and weights / bias folded by
        ``packing.fold_layernorm``.
    """
    g, ret = _gemm_build(a, w, bias, **kw)
    check(_lib.load().vidil_gemm(C.byref(g), _stream()), "gemm")
    return ret


def split_k_in_loop() -> bool:
    """True when the library forms the parity mode's compensated products inside one K loop (vidil_gemm_split_k_in_loop): such a
    consumer reads planes hi | lo of its [hi | lo | hi] operand rows only, so a producer may leave the third plane unwritten
    (``gemm(..., split3_out=, split3_planes=2)``)."""
    return bool(_lib.load().vidil_gemm_split_k_in_loop())


def split_k_serves(a, w, bias=None, **kw) -> bool:
    """Per CALL (vidil_gemm_split_k_serves): True when ``gemm(a, w, bias, split_k=True, **kw)`` would take the K-loop form and
    read planes hi | lo of ``a`` only.  A producer may leave the third plane of its rows unwritten only if this holds for EVERY
    consumer of those rows (process-wide ``split_k_in_loop()`` is necessary, not sufficient: the per-head epilogue with fewer
    than 8 tokens per sequence, or an unaligned vector, takes the plain K = 3 Kl product).  Launches nothing."""
    kw = dict(kw, split_k=True)
    kw.pop("a_planes", None)
    g, _ = _gemm_build(a, w, bias, **kw)
    rc = _lib.load().vidil_gemm_split_k_serves(C.byref(g))
    if rc < 0:
        check(rc, "gemm_split_k_serves")
    return rc == 1


def gemm_kernel_name(a, w, bias=None, **kw):
    """The kernel instantiation ``gemm`` would launch for these arguments, as rocprofv3 spells it."""
    g, _ = _gemm_build(a, w, bias, **kw)
    buf = C.create_string_buffer(128)
    check(_lib.load().vidil_gemm_kernel_name(C.byref(g), buf, 128), "gemm_kernel_name")
    return buf.value.decode()


def _gemm_build(a, w, bias=None, *, out=None, out_dtype=None, act=ACT_NONE, resid=None,
                heads=None, patch=None, arena=None, M=None, lda=None, out16=None, ln_stats_out=None, ln=None, w_scale=None,
                dtype16=None, rln=None, split3_out=None, split_k=False, split3_planes=3, a_planes=3):
    K_ = w.shape[1]
    if lda is None:
        M, Ka = a.shape
        if Ka != K_:
            raise VidilHipError(f"gemm: A is {tuple(a.shape)} but W is {tuple(w.shape)}")
    elif M is None:
        raise VidilHipError("gemm: strided A (lda) needs an explicit M")
    K = K_
    N = w.shape[0]
    g = GemmArgs()
    g.dtype = _dt(a, "gemm.A", fp8_ok=True)
    g.A = _ptr(a, a.dtype, "gemm.A")
    g.W = _ptr(w, a.dtype, "gemm.W")
    if rln is not None:                          # (gamma, beta, eps, stats of the residual): `resid` is a post-LN block's raw sum
        rg, rb, reps, rstats = rln
        g.rln_gamma, g.rln_beta = _ptr(rg, torch.float32, "gemm.rln_gamma"), _ptr(rb, torch.float32, "gemm.rln_beta")
        g.ln_stats, g.ln_eps = _ptr(rstats, torch.float32, "gemm.ln_stats"), float(reps)
    fp8 = g.dtype == DT_FP8
    # fp8 operands (tower mode): 16-bit outputs are written in the companion type (taken from the output buffers)
    t16 = a.dtype
    if fp8:
        t16 = dtype16
        for cand in ((heads or {}).get("q"), (heads or {}).get("k"), (heads or {}).get("vt")):
            if t16 is None and cand is not None:
                t16 = cand.dtype
        t16 = t16 or torch.float16
        g.dtype16 = _dt(t16, "gemm.dtype16")
        g.w_scale = _ptr(w_scale, torch.float32, "gemm.w_scale")
    g.bias = _ptr(bias, torch.float32, "gemm.bias")
    # A = [x_hi | x_lo | x_hi] rows, W = [W_hi | W_hi | W_lo] (the parity mode's operands); a_planes=2: the producer of ``a`` wrote
    # planes hi | lo only — the library then refuses (EINVAL) any launch that would read the third instead of the K-loop form
    if a_planes not in (2, 3) or (a_planes == 2 and not split_k):
        raise VidilHipError(f"gemm: a_planes={a_planes} (2 needs split_k=True)")
    g.split_k = (2 if a_planes == 2 else 1) if split_k else 0
    g.M, g.N, g.K = M, N, K
    g.lda = 0 if lda is None else lda
    g.act = act
    ret = None
    if heads is not None:
        g.epi = EPI_HEADS
        g.q = _ptr(heads.get("q"), t16, "gemm.q")
        g.k = _ptr(heads.get("k"), t16, "gemm.k")
        g.vt = _ptr(heads.get("vt"), t16, "gemm.vt")
        g.T, g.H = heads["T"], heads["H"]
        g.part0 = heads.get("part0", 0)
        g.t_off = heads.get("t_off", 0)
        g.Tq_cap = heads.get("Tq_cap", heads["T"])
        g.Tk_cap = heads.get("Tk_cap", heads["T"])
        g.NP = heads.get("NP", 0)
        g.q_scale = heads.get("q_scale", 1.0)
        g.kv_tiled = 1 if heads.get("tiled") else 0
    elif arena is not None:
        g.epi = EPI_ARENA
        g.q = _ptr(arena.get("q"), t16, "gemm.arena.q")
        g.k = _ptr(arena.get("k"), t16, "gemm.arena.k")
        g.vt = _ptr(arena.get("v"), t16, "gemm.arena.v")
        g.T, g.H = arena["T"], arena["H"]
        g.part0 = arena.get("part0", 0)
        g.t_off = arena.get("t_off", 0)
        g.Tk_cap = arena.get("Tcap", 0)
        g.arena_rows = arena.get("arena_rows", 0)
        g.slot_stride = arena.get("slot_stride", 1)
        g.q_scale = arena.get("q_scale", 1.0)
    elif patch is not None:
        g.epi = EPI_PATCH
        ret = patch["out"]
        g.out = _ptr(ret, torch.float32, "gemm.patch.out")
        g.ldo = ret.shape[-1]
        g.pos = _ptr(patch["pos"], torch.float32, "gemm.patch.pos")
        g.tpi = patch["tpi"]
    elif split3_out is not None:
        # f32 epilogue whose result leaves as [hi | lo | hi] operand rows of the next compensated GEMM (out16_split3); the f32
        # rows themselves are written only when ``out`` (f32) is given too
        if tuple(split3_out.shape) != (M, 3 * N) or split3_out.dtype != a.dtype or out16 is not None or resid is not None:
            raise VidilHipError(f"gemm: split3_out must be {a.dtype} [{M}, {3 * N}] (no out16 / resid beside it), got "
                                f"{split3_out.dtype} {tuple(split3_out.shape)}")
        g.epi = EPI_F32
        ret = split3_out
        if out is not None:
            g.out = _ptr(out, torch.float32, "gemm.out")
            g.ldo = out.shape[-1]
        else:
            g.ldo = N
        g.out16 = _ptr(split3_out, t16, "gemm.split3_out")
        g.ldo16 = 3 * N
        g.out16_split3 = 2 if split3_planes == 2 else 1      # (2: the consumer is a split_k launch in the K-loop form)
    else:
        if out is None:
            out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
        ret = out
        if out.dtype not in (a.dtype, torch.float32):
            raise VidilHipError(f"gemm: out must be {a.dtype} (the operand type) or float32, got {out.dtype}")
        g.epi = EPI_F32 if out.dtype == torch.float32 else (EPI_F8 if fp8 else EPI_F16)
        g.out = _ptr(out, None, "gemm.out")
        g.ldo = out.shape[-1]
        if resid is not None:
            if out.dtype != torch.float32:
                raise VidilHipError("gemm: resid needs an f32 output")
            g.resid = _ptr(resid, torch.float32, "gemm.resid")
    if out16 is not None:      # LN-fold producer: the raw stream in the operand type beside the f32 one
        g.out16 = _ptr(out16, t16, "gemm.out16")
        g.ldo16 = out16.shape[-1]
        if ln_stats_out is not None:     # + per-row (sum, sum of squares) partials, one pair per 64 output columns
            if tuple(ln_stats_out.shape) != (M, N // 64, 2):
                raise VidilHipError(f"gemm: ln_stats_out must be f32 [{M}, {N // 64}, 2], got {tuple(ln_stats_out.shape)}")
            g.ln_stats_out = _ptr(ln_stats_out, torch.float32, "gemm.ln_stats_out")
    if ln is not None:         # LN-fold consumer: (colsum f32 [N], eps, row partials f32 [M,K/64,2]); w / bias are the folded W', b'
        colsum, eps, stats = ln
        if tuple(stats.shape) != (M, K // 64, 2):
            raise VidilHipError(f"gemm: ln stats must be f32 [{M}, {K // 64}, 2], got {tuple(stats.shape)}")
        g.ln_fold = 1
        g.ln_colsum = _ptr(colsum, torch.float32, "gemm.ln_colsum")
        g.ln_stats = _ptr(stats, torch.float32, "gemm.ln_stats")
        g.ln_eps = float(eps)
    return g, ret


# ---------------------------------------------------------------------- row kernels
def layernorm(x, gamma, beta, eps, *, M=None, D=None, x_stride=None, out16=None, out32=None, split3=False, planes=3):
    """LayerNorm rows of f32 ``x``.  Rows are ``x_stride`` elements apart (default dense).
    split3: ``out16`` is [M, 3D] and receives the error-compensated operand rows [hi | lo | hi] (VIDIL_DT_SPLIT3); planes=2
    (VIDIL_DT_SPLIT2): only hi | lo are written — for consumers that are split_k GEMMs in the K-loop form (split_k_in_loop())."""
    lib = _lib.load()
    D = D if D is not None else x.shape[-1]
    M = M if M is not None else x.numel() // D
    x_stride = x_stride if x_stride is not None else D
    dt16 = DT_F16
    if out16 is not None:
        dt16 = _dt(out16, "ln.out16", fp8_ok=not split3) | (DT_SPLIT3 if split3 else 0) | (DT_SPLIT2 if split3 and planes == 2 else 0)
        if split3 and out16.shape[-1] != 3 * D:
            raise VidilHipError(f"layernorm: split3 out16 must be [M, {3 * D}], got {tuple(out16.shape)}")
    check(lib.vidil_layernorm(_ptr(x, torch.float32, "ln.x"), x_stride, _ptr(gamma, torch.float32, "ln.gamma"),
                              _ptr(beta, torch.float32, "ln.beta"), float(eps), M, D,
                              _ptr(out16, None, "ln.out16"), dt16,
                              _ptr(out32, torch.float32, "ln.out32"), _stream()), "layernorm")


def split3(x32, out16):
    """out16 [M,3D] = [hi | lo | hi] of f32 x [M,D] (hi = T16(x), lo = T16(x - hi)): error-compensated GEMM operand rows."""
    M, D = x32.shape
    check(_lib.load().vidil_split3_f32(_ptr(x32, torch.float32, "split3.x"), _ptr(out16, None, "split3.out"), M, D,
                                       _dt(out16, "split3.out"), _stream()), "split3")
    return out16


def attention(q, k, vt, out, *, Bq, H, Nq, Nk, Tq_cap, Tk_cap, NP, kv_group=1, causal=False,
              causal_off=0, kv_len=None, kv_index=None, group_start=None, max_group=0, ldo=None, kv_tiled=False,
              split3=False):
    """group_start: int32 [n_kv+1] device prefix table (query batches per kv batch), with max_group.
    kv_tiled: k / vt are fragment-tiled (gemm heads=dict(tiled=True)); at most 32 query rows per unit.
    split3: ``out`` is [rows, 3*H*64] and receives the error-compensated operand rows [hi | lo | hi] (VIDIL_DT_SPLIT3)."""
    lib = _lib.load()
    ldo = ldo if ldo is not None else (3 if split3 else 1) * H * 64
    n_kv = 0 if group_start is None else group_start.numel() - 1
    t16 = q.dtype
    check(lib.vidil_attention(_ptr(q, t16, "attn.q"), _ptr(k, t16, "attn.k"),
                              _ptr(vt, t16, "attn.vt"), _ptr(out, None, "attn.out"),
                              _ptr(kv_len, torch.int32, "attn.kv_len"), _ptr(kv_index, torch.int32, "attn.kv_index"),
                              _ptr(group_start, torch.int32, "attn.group_start"), n_kv, max_group,
                              Bq, H, Nq, Nk, Tq_cap, Tk_cap, NP,
                              kv_group, int(bool(causal)), causal_off, ldo, int(bool(kv_tiled)), _dt(q, "attn.q"),
                              _dt(out, "attn.out", fp8_ok=True) | (DT_SPLIT3 if split3 else 0), _stream()), "attention")
    return out


def _f32_view(t, what):
    """(data pointer, row stride in elements) of an f32 tensor whose LAST dim is contiguous and whose leading dims collapse to
    one row index (a [rows, C] matrix, or a column slice of one)."""
    from .packing import require_cuda
    require_cuda(t, what)
    if t.dtype != torch.float32 or t.stride(-1) != 1:
        raise VidilHipError(f"{what}: f32 tensor with a contiguous last dimension expected, got {t.dtype} strides {t.stride()}")
    t2 = t if t.dim() == 2 else t.flatten(0, -2)      # (raises if the leading dims do not collapse without a copy)
    if t2.data_ptr() != t.data_ptr():
        raise VidilHipError(f"{what}: leading dimensions do not collapse to rows without a copy")
    return t2.data_ptr(), t2.stride(0)


def attention_f32(q, k, v, out, *, Bq, H, Nq, Nk, kv_rows=None, kv_group=1, causal=False, causal_off=0, kv_len=None, kv_index=None,
                  group_start=None, max_group=0, scale=0.125, split3=None, anc=None, arena_rows=0, arith=0, kv16=False, planes=3):
    """softmax(q k^T * scale) v in f32 (the attention of the parity precision mode; vidil_attention_f32).

    arith: 0 = plain f32 arithmetic; 1 = split-operand 16-bit MFMA (every operand as hi + lo, three products per contraction:
    ~1e-6 of the logit scale from f32 arithmetic at a fifth of its cost; dense forms — the arena form always runs in f32).
    kv16 (arith 1): ``k`` / ``v`` are 16-bit FRAGMENT TILES [n_kv, H, kv_rows, 64] (project_cross_kv(tiled=True)) instead of f32
    rows — the decode steps' cross-attention, at most 32 query rows per unit; Q and the probabilities are split, K / V as stored.

    q, k, v: f32 matrices — typically COLUMN SLICES of the row-major output of a projection GEMM (``qkv32[:, :C]``, ``[:, C:2*C]``,
    ``[:, 2*C:]``): row stride and column offset are taken from the views, head h occupies columns h*64 .. h*64+63.
    q rows: query batch b at rows b*Nq ..; k / v rows: kv batch j at rows j*kv_rows .. (default kv_rows = Nk).
    out: f32 [Bq*Nq, H*64], or a 16-bit [Bq*Nq, 3*H*64] tensor receiving [hi | lo | hi] rows (split3 defaults to that case).
    anc (+ arena_rows): the arena form — Nq == 1, key j of query row b is row j*arena_rows + anc[b][j] of k / v."""
    a = _lib.AttnF32Args()
    a.q, a.ldq = _f32_view(q, "attention_f32.q")
    if kv16:
        if k.dtype != v.dtype or k.dtype != out.dtype:
            raise VidilHipError(f"attention_f32 (kv16): k / v / out must share one 16-bit dtype, got {k.dtype} / {v.dtype} / {out.dtype}")
        a.k, a.v = _ptr(k, None, "attention_f32.k"), _ptr(v, None, "attention_f32.v")
        _dt(k, "attention_f32.k")
        a.ldk = a.ldv = 0
    else:
        a.k, a.ldk = _f32_view(k, "attention_f32.k")
        a.v, a.ldv = _f32_view(v, "attention_f32.v")
    a.q_off = a.k_off = a.v_off = 0
    if split3 is None:
        split3 = out.dtype != torch.float32
    a.out = _ptr(out, None, "attention_f32.out")
    a.ldo = out.stride(-2) if out.dim() >= 2 else out.shape[-1]
    a.out_mode = (3 if planes == 2 else 2) if split3 else 0       # (3: [hi | lo | hi] rows with planes hi | lo written only)
    a.dtype16 = _dt(out, "attention_f32.out") if split3 else DT_F16
    a.Bq, a.H, a.Nq, a.Nk = Bq, H, Nq, Nk
    a.kv_rows = Nk if kv_rows is None else kv_rows
    a.kv_group = kv_group
    a.kv_index = _ptr(kv_index, torch.int32, "attention_f32.kv_index")
    a.group_start = _ptr(group_start, torch.int32, "attention_f32.group_start")
    a.n_kv = 0 if group_start is None else group_start.numel() - 1
    a.max_group = max_group
    a.kv_len = _ptr(kv_len, torch.int32, "attention_f32.kv_len")
    a.causal, a.causal_off = int(bool(causal)), causal_off
    a.anc = _ptr(anc, torch.int32, "attention_f32.anc")
    a.anc_ld = 0 if anc is None else anc.stride(0)
    a.arena_rows = arena_rows
    a.scale = float(scale)
    a.arith = 0 if anc is not None else int(arith)          # (the arena form has no split-operand kernel: f32 arithmetic)
    a.kv16 = 1 if kv16 else 0
    check(_lib.load().vidil_attention_f32(C.byref(a), _stream()), "attention_f32")
    return out


def resample_u8(src, dst, bounds, coeffs, *, vertical, src_row0=0):
    """One pass of Pillow's 8-bit resize: src u8 [B,in_h,in_w,3] -> dst u8 [B,out_h,out_w,3] (see vidil_resample_u8)."""
    B, in_h, in_w, _ = src.shape
    _, out_h, out_w, _ = dst.shape
    check(_lib.load().vidil_resample_u8(_ptr(src, torch.uint8, "resample.src"), _ptr(dst, torch.uint8, "resample.dst"),
                                        B, in_h, in_w, out_h, out_w, 1 if vertical else 0,
                                        _ptr(bounds, torch.int32, "resample.bounds"),
                                        _ptr(coeffs, torch.int32, "resample.coeffs"), coeffs.shape[1], src_row0,
                                        _stream()), "resample_u8")
    return dst


def patch_row_halfs(ps: int) -> int:
    """Columns of a patch row: 3*ps*ps rounded up to a multiple of 64 (zero padded; == 3*ps*ps when ps % 8 == 0)."""
    return (3 * ps * ps + 63) // 64 * 64


def patchify_f32(img, ps, out=None, dtype=torch.float16, split3=False):
    """split3: rows are [hi | lo | hi] of the f32 pixel values, 3 * patch_row_halfs(ps) wide (VIDIL_DT_SPLIT3)."""
    lib = _lib.load()
    B, Cc, S, S2 = img.shape
    if Cc != 3 or S != S2:
        raise VidilHipError(f"patchify_f32: expected [B,3,S,S], got {tuple(img.shape)}")
    G = S // ps
    if out is None:
        out = torch.empty((B * G * G, (3 if split3 else 1) * patch_row_halfs(ps)), dtype=dtype, device=img.device)
    check(lib.vidil_patchify_f32(_ptr(img, torch.float32, "patchify.img"), _ptr(out, None, "patchify.out"),
                                 B, S, ps, _dt(out, "patchify.out") | (DT_SPLIT3 if split3 else 0), _stream()), "patchify_f32")
    return out


def patchify_u8(img, ps, mean, std, out=None, dtype=torch.float16, split3=False):
    lib = _lib.load()
    B, S, S2, Cc = img.shape
    if Cc != 3 or S != S2:
        raise VidilHipError(f"patchify_u8: expected [B,S,S,3], got {tuple(img.shape)}")
    G = S // ps
    if out is None:
        out = torch.empty((B * G * G, (3 if split3 else 1) * patch_row_halfs(ps)), dtype=dtype, device=img.device)
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.vidil_patchify_u8(_ptr(img, torch.uint8, "patchify.img"), _ptr(out, None, "patchify.out"),
                                B, S, ps, m3, s3, _dt(out, "patchify.out") | (DT_SPLIT3 if split3 else 0), _stream()), "patchify_u8")
    return out


def set_cls_row(x, cls, pos0, B, T, D):
    check(_lib.load().vidil_set_cls_row(_ptr(x, torch.float32), _ptr(cls, torch.float32), _ptr(pos0, torch.float32),
                                        B, T, D, _stream()), "set_cls_row")


def embed_tokens(ids, word, pos, out, *, T, pos_off=0):
    M = ids.numel()
    D = word.shape[1]
    check(_lib.load().vidil_embed_tokens(_ptr(ids, torch.int32, "embed.ids"), _ptr(word, torch.float32),
                                         _ptr(pos, torch.float32), _ptr(out, torch.float32), M, T, pos_off, D,
                                         word.shape[0], _stream()), "embed_tokens")
    return out


def gather_rows(x, idx, out=None):
    n, D = idx.numel(), x.shape[-1]
    if out is None:
        out = torch.empty((n, D), dtype=torch.float32, device=x.device)
    check(_lib.load().vidil_gather_rows_f32(_ptr(x, torch.float32), _ptr(idx, torch.int32), _ptr(out, torch.float32),
                                            n, D, _stream()), "gather_rows")
    return out


def l2_normalize_rows(x):
    n, D = x.shape
    check(_lib.load().vidil_l2_normalize_rows(_ptr(x, torch.float32), n, D, _stream()), "l2_normalize_rows")
    return x


# ------------------------------------------------------------------------ beam search
def logsoftmax_topk(logits, beam_scores, B, nb, ban_token=-1, out_scores=None, out_index=None, beams_in_logits=None,
                    seqs=None, cur_len=0, penalty=1.0):
    """``seqs`` (i32 [B*nb, max_len], the beams' token ids) with ``cur_len`` and ``penalty``: the repetition penalty of
    ``generate(..., repetition_penalty=penalty)`` on the log-probabilities of the tokens each row already holds."""
    V = logits.shape[-1]
    dev = logits.device
    if out_scores is None:
        out_scores = torch.empty((B, 2 * nb), dtype=torch.float32, device=dev)
    if out_index is None:
        out_index = torch.empty((B, 2 * nb), dtype=torch.int32, device=dev)
    nbl = nb if beams_in_logits is None else beams_in_logits
    if seqs is not None:
        if seqs.dim() != 2 or seqs.shape[0] != B * nb or seqs.stride(1) != 1:
            raise ValueError(f"logsoftmax_topk: seqs must be [B*nb={B * nb}, max_len] with unit column stride, got {tuple(seqs.shape)}")
        check(_lib.load().vidil_logsoftmax_topk_penalty(_ptr(logits, torch.float32, "topk.logits"),
                                                        _ptr(beam_scores, torch.float32, "topk.beam_scores"), B, nb, nbl, V,
                                                        ban_token, _ptr(seqs, torch.int32, "topk.seqs"), cur_len, seqs.stride(0),
                                                        float(penalty), _ptr(out_scores, torch.float32),
                                                        _ptr(out_index, torch.int32), _stream()), "logsoftmax_topk_penalty")
        return out_scores, out_index
    check(_lib.load().vidil_logsoftmax_topk(_ptr(logits, torch.float32, "topk.logits"),
                                            _ptr(beam_scores, torch.float32, "topk.beam_scores"), B, nb, nbl, V,
                                            ban_token, _ptr(out_scores, torch.float32), _ptr(out_index, torch.int32),
                                            _stream()), "logsoftmax_topk")
    return out_scores, out_index


class BeamBuffers:
    """Device-resident beam-search state for ``B`` images x ``nb`` beams."""

    def __init__(self, B, nb, max_len, device):
        self.B, self.nb, self.max_len = B, nb, max_len
        i32 = dict(dtype=torch.int32, device=device)
        self.seqs = torch.zeros((B * nb, max_len), **i32)
        self.seqs_next = torch.zeros((B * nb, max_len), **i32)
        self.beam_scores = torch.zeros((B * nb,), dtype=torch.float32, device=device)
        self.beam_idx = torch.zeros((B * nb,), **i32)
        self.next_tok = torch.zeros((B * nb,), **i32)
        self.done = torch.zeros((B,), **i32)
        self.n_hyp = torch.zeros((B,), **i32)
        self.hyp_score = torch.zeros((B, nb), dtype=torch.float64, device=device)
        self.hyp_len = torch.zeros((B, nb), **i32)
        self.hyp_tok = torch.zeros((B, nb, max_len), **i32)
        self.worst = torch.full((B,), 1e9, dtype=torch.float64, device=device)
        self.n_done = torch.zeros((1,), **i32)

    def reset(self, prompt_ids):
        """prompt_ids: int32 [B, P]; beams of an image start identical, scores [0,-1e9,...]."""
        B, nb = self.B, self.nb
        P = prompt_ids.shape[1]
        if getattr(self, "swapped", False):   # every search starts from the same buffer orientation (captured
            self.swap()                        # decode-step graphs bake the pointers of each step in)
        self.seqs.zero_()
        self.seqs[:, :P] = prompt_ids.repeat_interleave(nb, dim=0)
        bs = torch.full((B, nb), -1e9, dtype=torch.float32, device=self.seqs.device)
        bs[:, 0] = 0.0
        self.beam_scores.copy_(bs.view(-1))
        self.done.zero_(); self.n_hyp.zero_(); self.worst.fill_(1e9); self.n_done.zero_()

    def struct(self):
        s = BeamState()
        for name, _ in BeamState._fields_:
            setattr(s, name, getattr(self, name).data_ptr())
        return s

    def swap(self):
        self.seqs, self.seqs_next = self.seqs_next, self.seqs
        self.swapped = not getattr(self, "swapped", False)


def beam_update(bufs: BeamBuffers, cand_scores, cand_index, V, cur_len, eos_id, pad_id):
    st = bufs.struct()
    check(_lib.load().vidil_beam_update(C.byref(st), _ptr(cand_scores, torch.float32), _ptr(cand_index, torch.int32),
                                        bufs.B, bufs.nb, V, cur_len, bufs.max_len, eos_id, pad_id, _stream()),
          "beam_update")
    bufs.swap()


def beam_finalize(bufs: BeamBuffers, cur_len, eos_id, pad_id):
    dev = bufs.seqs.device
    out_tok = torch.empty((bufs.B, bufs.max_len), dtype=torch.int32, device=dev)
    out_len = torch.empty((bufs.B,), dtype=torch.int32, device=dev)
    out_score = torch.empty((bufs.B,), dtype=torch.float32, device=dev)
    st = bufs.struct()
    check(_lib.load().vidil_beam_finalize(C.byref(st), bufs.B, bufs.nb, cur_len, bufs.max_len, eos_id, pad_id,
                                          _ptr(out_tok), _ptr(out_len), _ptr(out_score), _stream()), "beam_finalize")
    return out_tok, out_len, out_score


def sample_top_k_top_p(logits, seqs, done, n_done, next_tok, *, cur_len, min_length, eos_id, pad_id, top_k=50,
                       top_p=0.9, rep_penalty=1.1, seed=0, step=0, row_offset=0):
    """One nucleus-sampling step on logits f32 [B,V] (see vidil_sample_top_k_top_p); appends to seqs i32 [B,max_len]."""
    B, V = logits.shape
    check(_lib.load().vidil_sample_top_k_top_p(_ptr(logits, torch.float32, "sample.logits"), _ptr(seqs, torch.int32),
                                               _ptr(done, torch.int32), _ptr(n_done, torch.int32),
                                               _ptr(next_tok, torch.int32), B, V, seqs.shape[1], cur_len, min_length,
                                               eos_id, pad_id, top_k, float(top_p), float(rep_penalty), int(seed), step,
                                               row_offset, _stream()), "sample_top_k_top_p")


def beam_ancestry(anc_src, anc_dst, beam_idx, cur_pos):
    """anc_dst[r][:cur_pos] = anc_src[beam_idx[r]][:cur_pos]; anc_dst[r][cur_pos] = r  (i32 [rows,Tcap] tables)."""
    rows, Tcap = anc_src.shape
    check(_lib.load().vidil_beam_ancestry(_ptr(anc_src, torch.int32), _ptr(anc_dst, torch.int32),
                                          _ptr(beam_idx, torch.int32), rows, Tcap, cur_pos, _stream()), "beam_ancestry")


def beam_attention(q, k_arena, v_arena, anc, out, *, rows, H, n_keys, ldo=None, split3=False):
    """Decode-step self-attention over the KV arena: q f16 [rows,H*64]; arenas f16 [Tcap,arena_rows,H*64];
    anc i32 [rows,Tcap]; out f16 [rows,ldo] (split3: [rows, 3*H*64] = [hi | lo | hi], VIDIL_DT_SPLIT3)."""
    Tcap, arena_rows = k_arena.shape[0], k_arena.shape[1]
    t16 = q.dtype
    check(_lib.load().vidil_beam_attention(_ptr(q, t16), _ptr(k_arena, t16), _ptr(v_arena, t16), _ptr(anc, torch.int32),
                                           _ptr(out, t16), rows, H, n_keys, arena_rows, anc.shape[1],
                                           ldo if ldo is not None else out.shape[-1], _dt(q, "beam_attention.q"),
                                           _dt(q, "beam_attention.q") | (DT_SPLIT3 if split3 else 0), _stream()),
          "beam_attention")


def scan_scores(img, txt):
    """Exact-f32 dense scores img [NF,D] · txt [NC,D]^T -> f32 [NF,NC] (the chain of vidil_scan_topk)."""
    NF, D = img.shape
    NC = txt.shape[0]
    out = torch.empty((NF, NC), dtype=torch.float32, device=img.device)
    check(_lib.load().vidil_scan_scores(_ptr(img, torch.float32, "scores.img"), _ptr(txt, torch.float32, "scores.txt"), NF, D,
                                        NC, _ptr(out, torch.float32), _stream()), "scan_scores")
    return out


def topk_rows(x, k):
    """Sorted top-k of every row of f32 [R,N] (value desc, index asc): (values f32 [R,k], indices i32 [R,k])."""
    if x.dim() != 2 or x.stride(1) != 1:
        raise VidilHipError(f"topk_rows: expected a 2-D f32 tensor with contiguous rows, got {tuple(x.shape)} strides {x.stride()}")
    R, N = x.shape
    ov = torch.empty((R, k), dtype=torch.float32, device=x.device)
    oi = torch.empty((R, k), dtype=torch.int32, device=x.device)
    if x.dtype != torch.float32 or not x.is_cuda:
        raise VidilHipError("topk_rows: expected a float32 tensor on the GPU (there is no CPU fallback)")
    check(_lib.load().vidil_topk_rows(C.c_void_p(x.data_ptr()), x.stride(0), R, N, k, _ptr(ov, torch.float32),
                                      _ptr(oi, torch.int32), _stream()), "topk_rows")
    return ov, oi


# ------------------------------------------------------------------------- ontology scan
def scan_topk_ws_bytes(NF, NCpad, topk):
    return int(_lib.load().vidil_scan_topk_ws_bytes(NF, NCpad, topk))


def scan_topk(img, txt, seg_start, seg_len, topk, workspace=None):
    """img f32 [NF,D]; txt f32 [NCpad,D]; seg_start/seg_len: python lists per category."""
    NF, D = img.shape
    ncat = len(seg_start)
    dev = img.device
    need = scan_topk_ws_bytes(NF, txt.shape[0], topk)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=dev)
    out_i = torch.empty((NF, ncat, topk), dtype=torch.int32, device=dev)
    out_s = torch.empty((NF, ncat, topk), dtype=torch.float32, device=dev)
    ss = (C.c_int32 * ncat)(*seg_start)
    sl = (C.c_int32 * ncat)(*seg_len)
    check(_lib.load().vidil_scan_topk(_ptr(img, torch.float32, "scan.img"), _ptr(txt, torch.float32, "scan.txt"), NF, D,
                                      ncat, ss, sl, topk, _ptr(workspace), _ptr(out_i), _ptr(out_s), _stream()),
          "scan_topk")
    return out_i, out_s
