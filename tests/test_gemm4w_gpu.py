"""The two 256x256 GEMM kernels of the library — gemm4w.hip (four waves with 128x128 register blocks, one continuous K-tile
stream, lazily evaluated packed epilogue) and gemm256.hip (eight waves, staggered groups) — must be BIT-IDENTICAL: same k
order per output element, same epilogue arithmetic (gemm_epilogue.inc in its two forms), same LayerNorm statistics order —
for every epilogue of the hot path, both 16-bit operand types, ragged M / N, 2 to 48 K-tiles.  The dispatch
(vidil_gemm256_variant) may then pick either by speed alone; $VIDIL_GEMM4W = 0 / 1 forces one.
Reference ops: nn.Linear calls of models/vit.py:35-41,72,84 and models/med.py:153-171,236,301,314."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _k():
    from vidil_amd import kernels
    return kernels


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def both(k, fn, *a, **kw):
    """Run the same GEMM through gemm256 (VIDIL_GEMM4W=0) and through gemm4w (VIDIL_GEMM4W=1)."""
    old = os.environ.get("VIDIL_GEMM4W")
    try:
        os.environ["VIDIL_GEMM4W"] = "0"
        ref = fn()
        os.environ["VIDIL_GEMM4W"] = "1"
        got = fn()
    finally:
        if old is None:
            os.environ.pop("VIDIL_GEMM4W", None)
        else:
            os.environ["VIDIL_GEMM4W"] = old
    return ref, got


def _row_partials(x32):
    M, D = x32.shape
    xs = x32.view(M, D // 64, 64)
    return torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).contiguous()


_SLOW = pytest.mark.slow      # (the wide sweeps run under $VIDIL_RUN_SLOW=1 / -m "gpu and slow"; two shapes stay in -m gpu)
SHAPES = [pytest.param(256 * 300, 768, 128, marks=_SLOW), (256 * 8 + 5, 256 * 8, 192), pytest.param(197 * 130 + 37, 768, 768, marks=_SLOW),
          pytest.param(197 * 40 + 3, 3072, 768, marks=_SLOW), (197 * 130, 768, 3072), pytest.param(256 * 90, 832, 512, marks=_SLOW),
          pytest.param(197 * 70, 2304, 2304, marks=_SLOW)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm4w_16bit_and_f32_outputs_equal_gemm256(dtype, M, N, K):
    k = _k()
    a = _rand(M, K, seed=1).to(dtype).to(DEV)
    w = _rand(N, K, scale=0.05, seed=2).to(dtype).to(DEV)
    bias = _rand(N, seed=3).to(DEV)
    for act in (k.ACT_NONE, k.ACT_GELU_ERF, k.ACT_QUICK_GELU):
        ref, got = both(k, lambda: k.gemm(a, w, bias, out=torch.empty(M, N, dtype=dtype, device=DEV), act=act), a, w, bias, act=act)
        assert torch.equal(ref, got), (act, (ref.float() - got.float()).abs().max())
    x0 = _rand(M, N, seed=4).to(DEV)

    def run():
        x = x0.clone()
        k.gemm(a, w, bias, out=x, resid=x)
        return x
    ref, got = both(k, run, a, w, bias, out=x0, resid=x0)
    assert torch.equal(ref, got)
    # torch reference on a row sample (f32 accumulate over 16-bit operands)
    rows = torch.tensor([0, 1, 255, 256, M // 2, M - 2, M - 1], device=DEV)
    want = a[rows].float() @ w.float().t() + bias + x0[rows]
    assert torch.allclose(got[rows], want, rtol=1e-4, atol=2e-3 * (K / 768) ** 0.5)
    if N % 64 == 0:
        x16 = torch.zeros(M, N, dtype=dtype, device=DEV)
        st = torch.zeros(M, N // 64, 2, device=DEV)

        def run_stats():
            x = x0.clone()
            x16.zero_()
            st.zero_()
            k.gemm(a, w, bias, out=x, resid=x, out16=x16, ln_stats_out=st)
            return torch.cat([x.view(-1), x16.float().view(-1), st.view(-1)])
        ref, got = both(k, run_stats, a, w, bias, out=x0, resid=x0, out16=x16, ln_stats_out=st)
        assert torch.equal(ref, got)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm4w_layernorm_fold_consumers_and_residual_layernorm_equal_gemm256(dtype):
    from vidil_amd.packing import fold_layernorm

    k = _k()
    M, D, N = 197 * 60 + 11, 768, 3072
    x = _rand(M, D, seed=10) * 1.5
    x[:, 5] += 9.0
    x[::7] += 2.0
    g, bt = _rand(D, seed=11) * 0.2 + 1.0, _rand(D, seed=12) * 0.2
    wf, bf, cs = fold_layernorm(_rand(N, D, scale=0.03, seed=13), _rand(N, seed=14) * 0.1, g, bt, dtype)
    x16 = x.to(dtype).to(DEV)
    st = _row_partials(x).to(DEV)
    wf, bf, cs = wf.to(DEV), bf.to(DEV), cs.to(DEV)
    for act in (k.ACT_GELU_ERF, k.ACT_NONE):
        ref, got = both(k, lambda: k.gemm(x16, wf, bf, act=act, ln=(cs, 1e-6, st)), x16, wf, bf, act=act, ln=(cs, 1e-6, st))
        assert torch.equal(ref, got)
    # folded QKV with the per-head scatter: V row-major (towers), V^T and fragment-tiled K/V
    B, T, H = 60, 197, 12
    Mq = B * T
    wq, bq, csq = fold_layernorm(_rand(3 * D, D, scale=0.03, seed=15), _rand(3 * D, seed=16) * 0.1, g, bt, dtype)
    wq, bq, csq = wq.to(DEV), bq.to(DEV), csq.to(DEV)
    xq, stq = x16[:Mq].contiguous(), st[:Mq].contiguous()
    for layout in ("rowmajor", "vt", "tiled"):
        q = torch.zeros(B, H, T, 64, dtype=dtype, device=DEV)
        kk = torch.zeros(B, H, 224 if layout == "tiled" else T, 64, dtype=dtype, device=DEV)
        if layout == "rowmajor":
            v = torch.zeros(B, H, T, 64, dtype=dtype, device=DEV)
            hd = dict(q=q, k=kk, vt=v, T=T, H=H, part0=0, t_off=0, Tq_cap=T, Tk_cap=T, NP=0, q_scale=0.125)
        elif layout == "vt":
            v = torch.zeros(B, H, 64, 208, dtype=dtype, device=DEV)
            hd = dict(q=q, k=kk, vt=v, T=T, H=H, part0=0, t_off=0, Tq_cap=T, Tk_cap=T, NP=208, q_scale=0.125)
        else:
            v = torch.zeros(B, H, 224, 64, dtype=dtype, device=DEV)
            hd = dict(q=q, k=kk, vt=v, T=T, H=H, part0=0, t_off=0, Tq_cap=T, Tk_cap=224, tiled=True, q_scale=0.125)

        def run():
            q.zero_(); kk.zero_(); v.zero_()
            k.gemm(xq, wq, bq, heads=hd, ln=(csq, 1e-6, stq))
            return torch.cat([q.float().view(-1), kk.float().view(-1), v.float().view(-1)])
        ref, got = both(k, run, xq, wq, bq, heads=hd, ln=(csq, 1e-6, stq))
        assert torch.equal(ref, got), layout
    # residual LayerNorm epilogue (post-LN text stack)
    Nr, Kr = 768, 768
    Mr = 35 * 700 + 9
    u = _rand(Mr, Nr, seed=20) * 1.3
    u[:, 11] += 7.0
    gr, br = (_rand(Nr, seed=21) * 0.2 + 1.0).to(DEV), (_rand(Nr, seed=22) * 0.2).to(DEV)
    a = _rand(Mr, Kr, seed=23).to(dtype).to(DEV)
    w, b = _rand(Nr, Kr, scale=0.03, seed=24).to(dtype).to(DEV), (_rand(Nr, seed=25) * 0.1).to(DEV)
    st_in = _row_partials(u).to(DEV)
    xx = u.to(DEV)
    x16o = torch.zeros(Mr, Nr, dtype=dtype, device=DEV)
    sto = torch.zeros(Mr, Nr // 64, 2, device=DEV)

    def run_rln():
        xx.copy_(u)
        x16o.zero_(); sto.zero_()
        k.gemm(a, w, b, out=xx, resid=xx, out16=x16o, ln_stats_out=sto, rln=(gr, br, 1e-12, st_in))
        return torch.cat([xx.view(-1), x16o.float().view(-1), sto.view(-1)])
    ref, got = both(k, run_rln, a, w, b, out=xx, resid=xx, out16=x16o, ln_stats_out=sto, rln=(gr, br, 1e-12, st_in))
    assert torch.equal(ref, got)


def test_gemm4w_patch_epilogue_and_plain_heads_equal_gemm256():
    k = _k()
    B, P, D, Kp = 140, 196, 768, 768
    a = _rand(B * P, Kp, seed=30).half().to(DEV)
    w = _rand(D, Kp, scale=0.05, seed=31).half().to(DEV)
    bias = _rand(D, seed=32).to(DEV)
    pos = _rand(P + 1, D, seed=33).to(DEV)
    x = torch.zeros(B * (P + 1), D, device=DEV)

    def run():
        x.zero_()
        k.gemm(a, w, bias, patch=dict(out=x, pos=pos, tpi=P))
        return x.clone()
    ref, got = both(k, run, a, w, bias, patch=dict(out=x, pos=pos, tpi=P))
    assert torch.equal(ref, got)
    # un-folded cross K|V projection into fragment tiles (the captioner's image K/V)
    T, H = 197, 12
    Bk = 120
    ak = _rand(Bk * T, D, seed=34).half().to(DEV)
    wk = _rand(2 * D, D, scale=0.05, seed=35).half().to(DEV)
    bk = _rand(2 * D, seed=36).to(DEV)
    kk = torch.zeros(Bk, H, 224, 64, dtype=torch.float16, device=DEV)
    v = torch.zeros(Bk, H, 224, 64, dtype=torch.float16, device=DEV)
    hd = dict(k=kk, vt=v, T=T, H=H, part0=1, t_off=0, Tk_cap=224, tiled=True)

    def run_kv():
        kk.zero_(); v.zero_()
        k.gemm(ak, wk, bk, heads=hd)
        return torch.cat([kk.float().view(-1), v.float().view(-1)])
    ref, got = both(k, run_kv, ak, wk, bk, heads=hd)
    assert torch.equal(ref, got)


def test_gemm4w_repeated_launches_are_deterministic_and_race_free(monkeypatch):
    """The two wave groups hand stream elements over through one barrier per K-tile: 20 launches of a shape with many
    tiles per workgroup and an odd tile count give the same bits every time, and the right ones."""
    k = _k()
    M, N, K = 256 * 131 + 77, 1152, 768
    a = _rand(M, K, seed=40).half().to(DEV)
    w = _rand(N, K, scale=0.05, seed=41).half().to(DEV)
    bias = _rand(N, seed=42).to(DEV)
    monkeypatch.setenv("VIDIL_GEMM4W", "1")
    assert k.gemm_kernel_name(a, w, bias, act=k.ACT_GELU_ERF).startswith("gemm4w_kernel")
    first = k.gemm(a, w, bias, act=k.ACT_GELU_ERF)
    for _ in range(20):
        assert torch.equal(k.gemm(a, w, bias, act=k.ACT_GELU_ERF), first)
    want = torch.nn.functional.gelu(a[-300:].float() @ w.float().t() + bias)
    assert torch.allclose(first[-300:].float(), want, rtol=3e-3, atol=3e-3)


def test_gemm4w_rows_times_k_beyond_2_to_the_31(monkeypatch):
    """709,200 rows x K = 3072: M * K = 2.18e9 elements.  The 4-wave kernel addresses A as a uniform 64-bit row-panel base
    plus an unsigned 32-bit byte offset inside the panel (256 rows x 6 KiB here); rows on both sides of the 2^31-element
    line are checked against torch, and the whole output against gemm256."""
    k = _k()
    M, N, K = 197 * 3600, 256, 3072
    g = torch.Generator(device=DEV).manual_seed(5)
    a = (torch.randn(M, K, generator=g, device=DEV) * 0.5).half()
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).half()
    bias = torch.randn(N, generator=g, device=DEV)
    monkeypatch.setenv("VIDIL_GEMM4W", "1")
    assert k.gemm_kernel_name(a, w, bias).startswith("gemm4w_kernel")
    out = k.gemm(a, w, bias)
    for lo in (0, 349_000, 699_040, M - 300):           # 699,051 is the first row past 2^31 elements
        rows = slice(lo, lo + 300)
        want = a[rows].float() @ w.float().t() + bias
        assert torch.allclose(out[rows].float(), want, rtol=2e-3, atol=2e-2), lo
    monkeypatch.setenv("VIDIL_GEMM4W", "0")
    assert torch.equal(k.gemm(a, w, bias), out)


@pytest.mark.parametrize("D,N", [(1024, 4096), pytest.param(512, 2048, marks=_SLOW), pytest.param(320, 1280, marks=_SLOW)])
def test_gemm4w_layernorm_fold_at_other_widths_equals_gemm256(D, N):
    """ViT-L (1024: 16 row partials per row — every part slot of a half-wave in use), 512 and a width whose partial count
    (5) is not a multiple of 4 (the masked slots of the statistics exchange)."""
    from vidil_amd.packing import fold_layernorm

    k = _k()
    dtype = torch.bfloat16
    M = 256 * 150 + 33
    x = _rand(M, D, seed=50) * 1.5
    x[:, 3] += 6.0
    g, bt = _rand(D, seed=51) * 0.2 + 1.0, _rand(D, seed=52) * 0.2
    wf, bf, cs = fold_layernorm(_rand(N, D, scale=0.03, seed=53), _rand(N, seed=54) * 0.1, g, bt, dtype)
    x16 = x.to(dtype).to(DEV)
    st = _row_partials(x16.float().cpu()).to(DEV)
    wf, bf, cs = wf.to(DEV), bf.to(DEV), cs.to(DEV)
    ref, got = both(k, lambda: k.gemm(x16, wf, bf, act=k.ACT_GELU_ERF, ln=(cs, 1e-6, st)))
    assert torch.equal(ref, got)
    want = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x16[:500].float().cpu(), (D,), g, bt, 1e-6) @ _rand(N, D, scale=0.03, seed=53).t() + _rand(N, seed=54) * 0.1)
    assert torch.allclose(got[:500].float().cpu(), want, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(10752, 768, 768), pytest.param(10752, 768, 3072, marks=_SLOW), pytest.param(128 * 70 + 5, 768, 192, marks=_SLOW), (9999, 832, 512)])
def test_gemm4w_128_row_tile_form_equals_the_other_kernels(monkeypatch, dtype, M, N, K):
    """The 128 x 256-tile form of gemm4w (mid-size grids: the decode steps' projections and FFN at ~10^4 beam rows) against
    whatever serves the problem without it (the small-tile kernel or gemm256 — themselves bit-identical): f32 + residual,
    16-bit with and without activations, the per-head Q scatter; ragged M and N."""
    k = _k()
    a = _rand(M, K, seed=1).to(dtype).to(DEV)
    w = _rand(N, K, scale=0.05, seed=2).to(dtype).to(DEV)
    bias = _rand(N, seed=3).to(DEV)
    x0 = _rand(M, N, seed=4).to(DEV)
    H = 12

    def run_all():
        outs = []
        x = x0.clone()
        k.gemm(a, w, bias, out=x, resid=x)
        outs.append(x)
        for act in (k.ACT_NONE, k.ACT_GELU_ERF, k.ACT_QUICK_GELU):
            outs.append(k.gemm(a, w, bias, act=act).float())
        outs.append(k.gemm(a, w, None).float())
        if N == H * 64:
            q = torch.zeros(M, H, 1, 64, dtype=dtype, device=DEV)
            k.gemm(a, w, bias, heads=dict(q=q, T=1, H=H, part0=0, t_off=0, Tq_cap=1, q_scale=0.125))
            outs.append(q.float().view(M, -1))
        return outs

    monkeypatch.setenv("VIDIL_GEMM4W128", "0")
    assert "gemm4w_kernel" not in k.gemm_kernel_name(a, w, bias, out=x0, resid=x0) or not k.gemm_kernel_name(a, w, bias, out=x0, resid=x0).endswith(", 2, false>")
    ref = run_all()
    monkeypatch.setenv("VIDIL_GEMM4W128", "1")
    assert k.gemm_kernel_name(a, w, bias, out=x0, resid=x0).startswith("gemm4w_kernel") and k.gemm_kernel_name(a, w, bias, out=x0, resid=x0).endswith(", 2, false>")
    got = run_all()
    for r, g in zip(ref, got):
        assert torch.equal(r, g)
    want = a[-200:].float() @ w.float().t() + bias + x0[-200:]
    assert torch.allclose(got[0][-200:], want, rtol=1e-4, atol=2e-3 * (K / 768) ** 0.5 + 1e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,stride", [(1, 1), (4, 3)])
def test_arena_epilogue_on_the_256_row_kernels(monkeypatch, dtype, T, stride):
    """EPI_ARENA (Q rows + K / V rows appended to the beam-search KV arena: include/vidil_hip.h) through gemm256, gemm4w
    and gemm4w's 128-row form — the decode steps' self-attention Q|K|V projection at ~10^4 beam rows used to run on the
    small-tile kernel only.  Expected values: the f32 GEMM (+ bias) rounded as the epilogue rounds (Q scaled by q_scale)."""
    k = _k()
    H, D = 12, 768
    nseq = 10752 // T
    M = nseq * T
    a = _rand(M, D, seed=1).to(dtype).to(DEV)
    w = _rand(3 * D, D, scale=0.05, seed=2).to(dtype).to(DEV)
    bias = _rand(3 * D, seed=3).to(DEV)
    t_off, Tcap = 5, 16
    rows = (nseq - 1) * stride + 1
    f32 = k.gemm(a, w, bias, out_dtype=torch.float32)
    qs = 0.125
    want_q = (f32[:, :D] * qs).to(dtype)
    want_k, want_v = f32[:, D:2 * D].to(dtype), f32[:, 2 * D:].to(dtype)
    names = set()
    for env in (dict(VIDIL_GEMM4W="0", VIDIL_GEMM4W128="0"), dict(VIDIL_GEMM4W="1", VIDIL_GEMM4W128="0"), dict(VIDIL_GEMM4W128="1")):
        for kk in ("VIDIL_GEMM4W", "VIDIL_GEMM4W128"):
            monkeypatch.delenv(kk, raising=False)
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        q = torch.zeros(M, D, dtype=dtype, device=DEV)
        ka = torch.zeros(Tcap, rows, D, dtype=dtype, device=DEV)
        va = torch.zeros(Tcap, rows, D, dtype=dtype, device=DEV)
        ar = dict(q=q, k=ka, v=va, T=T, H=H, part0=0, t_off=t_off, arena_rows=rows, slot_stride=stride, Tcap=Tcap, q_scale=qs)
        nm = k.gemm_kernel_name(a, w, bias, arena=ar)
        # (kernel + its row-tile count: gemm4w's template arguments end "..., <TM>, <C3>>", gemm256's have neither)
        names.add(nm.split("<")[0] + (nm.split(",")[-2].strip() if nm.startswith("gemm4w") else ""))
        k.gemm(a, w, bias, arena=ar)
        assert torch.equal(q, want_q)
        got_k = ka[t_off:t_off + T, ::stride].permute(1, 0, 2).reshape(M, D)
        got_v = va[t_off:t_off + T, ::stride].permute(1, 0, 2).reshape(M, D)
        assert torch.equal(got_k, want_k) and torch.equal(got_v, want_v)
        assert ka[:t_off].abs().sum() == 0 and ka[t_off + T:].abs().sum() == 0
    assert len(names) == 3, names


@pytest.mark.parametrize("dt16", [torch.float16, torch.bfloat16])
def test_gemm4w_fp8_operands_equal_gemm256(dt16):
    """Round 4: e4m3 operands on the 4-wave main loop (v_mfma_scale_f32_32x32x64_f8f6f4, K-tiles of 128) — the fp8 tower mode's
    three epilogues (fp8 hand-over with GELU, per-head scatter into the 16-bit companion type, f32 + residual) through both
    256-row kernels, bit for bit; ragged M, and the dispatch picks the 4-wave kernel at the tower's size."""
    from vidil_amd.packing import w8

    k = _k()
    F8 = torch.float8_e4m3fn
    B, T, H, D = 130, 197, 12, 768                  # 25,610 rows: 101 row tiles (the last one ragged) x 9 / 12 / 3 column tiles
    M = B * T
    a8 = _rand(M, D, seed=80).to(F8).to(DEV)
    # ---- per-head scatter
    wq, sq = w8(_rand(3 * D, D, scale=0.03, seed=81))
    bq = (_rand(3 * D, seed=82) * 0.1).to(DEV)

    def run_heads():
        q = torch.zeros(B, H, T, 64, dtype=dt16, device=DEV)
        kk, v = torch.zeros_like(q), torch.zeros_like(q)
        k.gemm(a8, wq.to(DEV), bq, w_scale=sq.to(DEV),
               heads=dict(q=q, k=kk, vt=v, T=T, H=H, part0=0, t_off=0, Tq_cap=T, Tk_cap=T, NP=0, q_scale=0.125))
        return q, kk, v

    ref, got = both(k, run_heads)
    for r, g in zip(ref, got):
        assert torch.equal(r.view(torch.int16), g.view(torch.int16))
    # ---- fc1: GELU, fp8 out
    w1, s1 = w8(_rand(3072, D, scale=0.03, seed=83))
    b1 = (_rand(3072, seed=84) * 0.1).to(DEV)

    def run_fc1():
        hid = torch.zeros(M, 3072, dtype=F8, device=DEV)
        k.gemm(a8, w1.to(DEV), b1, out=hid, act=k.ACT_GELU_ERF, w_scale=s1.to(DEV), dtype16=dt16)
        return hid

    ref, got = both(k, run_fc1)
    assert torch.equal(ref.view(torch.uint8), got.view(torch.uint8))
    # ---- fc2: f32 + residual (K = 3072)
    w2, s2 = w8(_rand(D, 3072, scale=0.03, seed=85))
    b2 = (_rand(D, seed=86) * 0.1).to(DEV)
    x0 = _rand(M, D, seed=87).to(DEV)
    hid = got

    def run_fc2():
        x = x0.clone()
        k.gemm(hid, w2.to(DEV), b2, out=x, resid=x, w_scale=s2.to(DEV), dtype16=dt16)
        return x

    ref, got = both(k, run_fc2)
    assert torch.equal(ref, got)
    # the dispatch: at the tower's size the 4-wave kernel serves the fp8 GEMMs, small grids stay on the 8-wave kernel
    big = torch.empty(197 * 512, D, dtype=F8, device=DEV)
    name = k.gemm_kernel_name(big, w1.to(DEV), b1, out=torch.empty(197 * 512, 3072, dtype=F8, device=DEV), act=k.ACT_GELU_ERF,
                              w_scale=s1.to(DEV), dtype16=dt16)
    assert name.startswith("gemm4w_kernel<fp8") or "gemm4w" in name, name
    name = k.gemm_kernel_name(a8[:197 * 8], w1.to(DEV), b1, out=torch.empty(197 * 8, 3072, dtype=F8, device=DEV), act=k.ACT_GELU_ERF,
                              w_scale=s1.to(DEV), dtype16=dt16)
    assert "gemm256" in name, name
