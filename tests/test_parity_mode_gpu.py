"""The "parity" precision mode (vidil_amd.packing.set_parity_mode, $VIDIL_PARITY): every GEMM of the caption path on
error-compensated operands — activations as [hi | lo | hi] rows (VIDIL_DT_SPLIT3 outputs / vidil_split3_f32), weights as
[W_hi | W_hi | W_lo], K tripled — so that BASELINE's "caption logits within 1e-3" holds as an ABSOLUTE bound against the
fp32 CPU oracle (reference: models/med.py:501-545,830-930 logits of BertLMHeadModel; models/vit.py:180-194).

Kernel level: each split3 producer equals vidil_split3_f32 of what the kernel writes in f32 / the plain kernel's own
rounding; one K-tripled GEMM reproduces the fp32 product to ~1e-6.  Model level: all 16 teacher-forced forward passes of
a beam search, asserted with abs_max = 1e-3 (and no relative reading)."""
import numpy as np
import pytest
import torch

from common import perturb_, synthetic_frames

pytestmark = pytest.mark.gpu
DEV = "cuda"
ABS_TOL = 1e-3            # BASELINE.json north_star: "caption logits within 1e-3 fp16"


def _k():
    from vidil_amd import kernels
    return kernels


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _split3_ref(x32, dtype):
    hi = x32.to(dtype)
    lo = (x32 - hi.float()).to(dtype)
    return torch.cat([hi, lo, hi], dim=-1)


def _join(x3):
    """[.., 3D] split rows -> the f32 value hi + lo they encode."""
    D = x3.shape[-1] // 3
    assert torch.equal(x3[..., :D], x3[..., 2 * D:])
    return x3[..., :D].float() + x3[..., D:2 * D].float()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_layernorm_split3_rows_equal_split3_of_its_f32_output(dtype):
    k = _k()
    M, D = 333, 768
    x = (_rand(M, D, seed=1) * 3 + 0.5).to(DEV)
    g, b = (_rand(D, seed=2) * 0.2 + 1).to(DEV), (_rand(D, seed=3) * 0.1).to(DEV)
    y32 = torch.empty(M, D, dtype=torch.float32, device=DEV)
    y3 = torch.empty(M, 3 * D, dtype=dtype, device=DEV)
    k.layernorm(x, g, b, 1e-6, out16=y3, out32=y32, split3=True)
    assert torch.equal(y3.cpu(), _split3_ref(y32.cpu(), dtype))
    # and it is what the stand-alone split kernel makes of the f32 rows
    assert torch.equal(k.split3(y32, torch.empty_like(y3)), y3)
    ref = torch.nn.functional.layer_norm(x.cpu(), (D,), g.cpu(), b.cpu(), 1e-6)
    assert (_join(y3.cpu()) - ref).abs().max().item() < (3e-6 if dtype == torch.float16 else 5e-5)


@pytest.mark.parametrize("ps,S", [(16, 64), (14, 56)])
def test_patchify_split3_rows_hold_the_f32_pixel_values(ps, S):
    k = _k()
    B = 3
    rng = np.random.default_rng(5)
    u8 = torch.from_numpy(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8))
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    p16 = k.patchify_u8(u8.to(DEV), ps, mean, std)
    p3 = k.patchify_u8(u8.to(DEV), ps, mean, std, split3=True)
    ldk = p16.shape[1]
    assert p3.shape == (p16.shape[0], 3 * ldk)
    assert torch.equal(p3[:, :ldk], p16) and torch.equal(p3[:, 2 * ldk:], p16)
    x = (u8.float() / 255.0 - torch.tensor(mean)) / torch.tensor(std)               # [B,S,S,3]
    G = S // ps
    ref = x.view(B, G, ps, G, ps, 3).permute(0, 1, 3, 5, 2, 4).reshape(B * G * G, 3 * ps * ps)
    got = _join(p3.cpu())
    assert (got[:, :3 * ps * ps] - ref).abs().max().item() < 2e-6
    assert (got[:, 3 * ps * ps:] == 0).all()
    # the f32 entry point too
    img = x.permute(0, 3, 1, 2).contiguous()
    q3 = k.patchify_f32(img.to(DEV), ps, split3=True)
    assert (_join(q3.cpu())[:, :3 * ps * ps] - ref).abs().max().item() < 1e-6


@pytest.mark.parametrize("Bq,H,Nq,Nk,kv_group,causal", [
    (3, 12, 197, 197, 1, False),     # staged kernel (ViT)
    (6, 12, 3, 197, 3, False),       # direct kernel (decode cross-attention)
    (2, 12, 4, 4, 1, True),          # one-wave kernel (prompt pass)
    (6, 4, 35, 197, 3, False),       # staged kernel, grouped queries
])
def test_attention_split3_output_carries_the_f32_result(Bq, H, Nq, Nk, kv_group, causal):
    k = _k()
    Bk = Bq // kv_group
    NP = (Nk + 15) // 16 * 16
    q = (_rand(Bq, H, Nq, 64, seed=30) * 0.125).half()
    kk = _rand(Bk, H, Nk, 64, seed=31).half()
    v = _rand(Bk, H, Nk, 64, seed=32).half()
    vt = torch.zeros((Bk, H, 64, NP), dtype=torch.float16)
    vt[..., k.vt_columns(Nk)] = v.transpose(-1, -2)
    C = H * 64
    args = dict(Bq=Bq, H=H, Nq=Nq, Nk=Nk, Tq_cap=Nq, Tk_cap=Nk, NP=NP, kv_group=kv_group, causal=causal)
    o16 = torch.zeros(Bq * Nq, C, dtype=torch.float16, device=DEV)
    o3 = torch.zeros(Bq * Nq, 3 * C, dtype=torch.float16, device=DEV)
    k.attention(q.to(DEV), kk.to(DEV), vt.to(DEV), o16, **args)
    k.attention(q.to(DEV), kk.to(DEV), vt.to(DEV), o3, split3=True, **args)
    # the hi plane is the (pinned) f16 rounding of the result; the plain kernels may round the final scale and the
    # conversion in one fused step, so allow one unit in the last place between the two
    assert torch.equal(o3[:, :C], o3[:, 2 * C:])
    assert torch.allclose(o3[:, :C].float(), o16.float(), rtol=1.1e-3, atol=1e-7)
    # hi + lo is closer to the exact attention of these (16-bit) operands than hi alone by orders of magnitude
    kr, vr = kk.float().repeat_interleave(kv_group, 0), v.float().repeat_interleave(kv_group, 0)
    s = q.float() @ kr.transpose(-1, -2)
    if causal:
        s = s.masked_fill(torch.arange(Nk)[None, :] > torch.arange(Nq)[:, None], float("-inf"))
    ref = (torch.softmax(s.double(), -1) @ vr.double()).permute(0, 2, 1, 3).reshape(Bq * Nq, C)
    e_hi = (o16.cpu().double() - ref).abs().max().item()
    e_split = (_join(o3.cpu()).double() - ref).abs().max().item()
    # (what remains is the f16 rounding of the probabilities inside the kernel — common to both outputs; where it dominates,
    #  e.g. three query rows over 197 keys in the direct kernel: 2.7e-4 vs 2.3e-4, the two errors are the same size and which
    #  one is smaller is a coin toss, so the comparison carries that much slack)
    assert e_split < 4e-4 and e_split <= e_hi + 5e-5, (e_split, e_hi)      # (slack cut to the measured coin-toss range, 4e-5)


def test_beam_attention_split3_output():
    k = _k()
    rows, H, n_keys, Tcap = 33, 12, 9, 20
    C = H * 64
    g = torch.Generator().manual_seed(7)
    q = (_rand(rows, C, seed=66) * 0.125).half()
    ka, va = _rand(Tcap, rows, C, seed=67).half(), _rand(Tcap, rows, C, seed=68).half()
    anc = torch.randint(0, rows, (rows, Tcap), generator=g, dtype=torch.int32)
    o16 = torch.zeros(rows, C, dtype=torch.float16, device=DEV)
    o3 = torch.zeros(rows, 3 * C, dtype=torch.float16, device=DEV)
    k.beam_attention(q.to(DEV), ka.to(DEV), va.to(DEV), anc.to(DEV), o16, rows=rows, H=H, n_keys=n_keys)
    k.beam_attention(q.to(DEV), ka.to(DEV), va.to(DEV), anc.to(DEV), o3, rows=rows, H=H, n_keys=n_keys, split3=True)
    assert torch.equal(o3[:, :C], o16) and torch.equal(o3[:, 2 * C:], o16)
    t = torch.arange(n_keys)
    kg = ka[t[None, :], anc[:, :n_keys].long()].double().view(rows, n_keys, H, 64)
    vg = va[t[None, :], anc[:, :n_keys].long()].double().view(rows, n_keys, H, 64)
    s = torch.einsum("rhd,rthd->rht", q.double().view(rows, H, 64), kg)
    ref = torch.einsum("rht,rthd->rhd", torch.softmax(s, dim=-1), vg).reshape(rows, C)
    assert (_join(o3.cpu()).double() - ref).abs().max().item() < 2e-6       # (f32 arithmetic throughout this kernel)


@pytest.mark.parametrize("M,N,K", [(197 * 3, 2304, 768), (12, 30524, 768), (197 * 130, 768, 3072)])
def test_split_operand_gemm_reproduces_the_fp32_product(M, N, K):
    """x_hi·W_hi + x_lo·W_hi + x_hi·W_lo in one K-tripled GEMM (small-tile and 256x256 kernels)."""
    from vidil_amd.packing import w3

    k = _k()
    x = _rand(M, K, seed=40)
    w = _rand(N, K, scale=0.03, seed=41)
    bias = _rand(N, seed=42)
    ref = (x.double() @ w.double().t() + bias.double())
    a3 = k.split3(x.to(DEV), torch.empty(M, 3 * K, dtype=torch.float16, device=DEV))
    got = k.gemm(a3, w3(w, dtype=torch.float16).to(DEV), bias.to(DEV), out_dtype=torch.float32).cpu().double()
    plain = k.gemm(x.half().to(DEV), w.half().to(DEV), bias.to(DEV), out_dtype=torch.float32).cpu().double()
    e, e_plain = (got - ref).abs().max().item(), (plain - ref).abs().max().item()
    print(f"M={M} N={N} K={K}: split-operand GEMM max|d| {e:.2e} (plain f16 operands {e_plain:.2e})")
    assert e < 1e-5 * max(1.0, ref.abs().max().item()) + 1e-5 and e < e_plain / 20


# =============================================================== model level: caption logits within 1e-3, ABSOLUTE
@pytest.fixture(scope="module")
def parity_captioner():
    from vidil_amd.blip import BLIP_Decoder
    from vidil_amd.packing import set_compute_dtype, set_parity_mode
    from vidil_amd.tokenizer import SyntheticBertTokenizer

    torch.manual_seed(0)
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=SyntheticBertTokenizer()).eval()
    perturb_(cap, 100)
    sd = {k: v.clone() for k, v in cap.state_dict().items()}
    cap = cap.to(DEV)
    set_compute_dtype("f16", cap)
    set_parity_mode(True, cap)
    return cap, sd


def test_parity_mode_vit_output_vs_oracle(parity_captioner):
    from oracle import clip_ref, vit_ref

    cap, sd = parity_captioner
    u8 = synthetic_frames(1, 3)[0]
    with torch.no_grad():
        y_ref = vit_ref.vit_forward(sd, clip_ref.preprocess_u8(u8))
    y32, y3 = cap.visual_encoder.forward_u8(torch.from_numpy(u8).to(DEV), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
    d = (y32.cpu() - y_ref).abs()
    print(f"parity-mode ViT-B/16 output vs fp32 oracle: max {d.max().item():.2e} mean {d.mean().item():.2e}")
    assert d.max().item() < 5e-5 and d.mean().item() < 5e-6          # (plain f16 operands: 1e-2 / 1e-3; 16-bit attention in the mode: 2.4e-4 / 2.1e-5)
    assert y3.shape == (3 * 197, 3 * 768)
    # (the model writes planes hi | lo only where every consumer is a split_k GEMM in the K-loop form: the third plane is unspecified)
    y3c = y3.cpu()
    assert (y3c[:, :768].float() + y3c[:, 768:1536].float() - y32.cpu().view(-1, 768)).abs().max().item() < 1e-6


def test_parity_mode_caption_logits_within_1e_3_absolute_on_every_forward_pass(parity_captioner):
    """BASELINE: "caption logits within 1e-3 fp16" — ViT + cross K/V + 12 decoder layers + LM head on the device, all 16
    forward passes of a beam search (prompt pass + 15 cached steps) teacher-forced with the fp32 oracle's own beam
    decisions, max|logit_hip - logit_ref| <= 1e-3 with NO scaling by the logit magnitude."""
    from oracle import beam_ref, clip_ref, med_ref, vit_ref
    from vidil_amd.blip import DecoderSession

    cap, sd = parity_captioner
    B, nb = 3, 3
    u8 = synthetic_frames(1, B)[0]
    with torch.no_grad():
        y_ref = vit_ref.vit_forward(sd, clip_ref.preprocess_u8(u8))
    enc3 = y_ref.repeat_interleave(nb, dim=0)
    state, otrace, calls = {}, [], []

    def step(ids, beam_idx):
        calls.append((ids.copy(), None if beam_idx is None else beam_idx.copy()))
        with torch.no_grad():
            past = None if beam_idx is None else med_ref.reorder_cache(state["cache"], torch.from_numpy(beam_idx))
            lg, state["cache"] = med_ref.decoder_logits(sd, torch.from_numpy(ids), enc3, past)
        return lg.numpy()

    prompt = cap.prompt_ids(B, "cpu").long().numpy()
    beam_ref.beam_search(step, prompt, num_beams=nb, max_length=20, min_length=5, eos_token_id=102, pad_token_id=0, trace=otrace)
    assert len(otrace) == 16
    _, y3 = cap.visual_encoder.forward_u8(torch.from_numpy(u8).to(DEV), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
    sess = DecoderSession(cap.text_decoder, y3, B, nb, 20)
    worst = 0.0
    table = []
    for s, (ids, beam_idx) in enumerate(calls):
        if s == 0:
            lg = sess.prefill(torch.from_numpy(ids).to(torch.int32).reshape(-1).to(DEV), ids.shape[1])
        else:
            lg = sess.step(torch.from_numpy(ids[:, -1].copy()).to(torch.int32).to(DEV),
                           torch.from_numpy(beam_idx).to(torch.int32).to(DEV), ids.shape[1] - 1)
        ref = torch.from_numpy(otrace[s]["logits"])
        d = (lg.cpu() - ref).abs()
        table.append(dict(step=s, max_abs=d.max().item(), mean_abs=d.mean().item(), ref_absmax=ref.abs().max().item()))
        worst = max(worst, d.max().item())
    import json
    import os

    from common import ROOT
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "logit_error_table_parity.json"), "w") as f:
        json.dump(table, f, indent=1)
    print(f"parity mode, 16 forward passes: worst max|d| = {worst:.3e} absolute (asserted <= {ABS_TOL:g}); "
          f"per pass: {', '.join('%.1e' % r['max_abs'] for r in table)}")
    assert worst <= ABS_TOL, table
    # free-running: the device beam search in parity mode produces the oracle's captions
    out_tok, _ = cap.generate_ids(y3, B, num_beams=nb, max_length=20, min_length=5)
    seqs, _ = beam_ref.beam_search(step, prompt, num_beams=nb, max_length=20, min_length=5, eos_token_id=102, pad_token_id=0)
    toks = out_tok.cpu().numpy()
    same = [bool(np.array_equal(toks[b][: len(seqs[b])], seqs[b])) for b in range(B)]
    # a search may legitimately take another branch where the ORACLE's own candidates are closer than the device's error
    # (random-init weights: near-flat distributions, f32 ties do occur): decisive = every adjacent candidate gap > 2e-5, twice
    # the mode's worst logit error on this sample (8e-6; ADVICE r4: the mask was 1e-4)
    gaps = np.stack([np.min(t["cand_scores"][:, :-1] - t["cand_scores"][:, 1:], axis=1) for t in otrace]).min(axis=0)
    print(f"parity mode free-running captions equal to the fp32 oracle: {sum(same)}/{B}; smallest candidate gap per image "
          f"{', '.join('%.1e' % g for g in gaps)}")
    for b in range(B):
        assert same[b] or gaps[b] < 2e-5, (b, gaps[b])
    assert sum(same) >= B - 1


def test_parity_mode_is_refused_with_fp8_and_off_by_default():
    from vidil_amd.packing import parity_mode, set_compute_dtype, set_parity_mode
    from vidil_amd.vit import VisionTransformer

    v = VisionTransformer(img_size=32, patch_size=16, embed_dim=256, depth=1, num_heads=4)
    assert parity_mode(v) is False
    set_parity_mode(True, v)
    set_compute_dtype("fp8", v)
    with pytest.raises(ValueError):
        v.parity


def test_parity_mode_itm_logits_and_capfilt_decisions_vs_oracle():
    """The filter side of CapFilt in the parity precision mode (BLIP_ITM.itm_pairs takes the whole-stack route on
    error-compensated operands): ITM logits of (frame, caption) pairs with padding against the fp32 oracle
    (models/blip_itm.py:41-58, models/med.py BertModel) — 1e-4 where the plain f16 path is asserted at 2e-3 — and a small
    CapFiltEngine run whose captions and kept lists equal the oracle pipeline's."""
    from oracle import clip_ref, med_ref, pipeline_ref, vit_ref
    from vidil_amd.blip import BLIP_Decoder
    from vidil_amd.blip_itm import BLIP_ITM
    from vidil_amd.capfilt import CapFiltEngine
    from vidil_amd.packing import set_compute_dtype, set_parity_mode
    from vidil_amd.tokenizer import SyntheticBertTokenizer

    torch.manual_seed(0)
    tok = SyntheticBertTokenizer()
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=tok).eval()
    itm = BLIP_ITM(image_size=224, vit="base", tokenizer=tok).eval()
    perturb_(cap, 100)
    perturb_(itm, 101)
    sd_cap = {k: v.clone() for k, v in cap.state_dict().items()}
    sd_itm = {k: v.clone() for k, v in itm.state_dict().items()}
    cap, itm = cap.to(DEV), itm.to(DEV)
    set_compute_dtype("f16", cap, itm)
    set_parity_mode(True, cap, itm)
    # ---- ITM logits of 4 frames x their captions (different lengths -> padding) through the reference call shape
    F = 4
    u8 = synthetic_frames(1, F, first_video=3)[0]
    x = clip_ref.preprocess_u8(u8)
    captions = ["w2000 w2001 w2002", "w3000 w3001 w3002 w3003 w3004 w3005 w3006", "a picture of w4000", "w5000 " * 12]
    got = itm(x.to(DEV), captions, match_head="itm").cpu()
    with torch.no_grad():
        y_ref = vit_ref.vit_forward(sd_itm, x)
        ids, lens = itm.tokenize(captions)
        ref = med_ref.itm_logits(sd_itm, y_ref, ids.long(), (torch.arange(ids.shape[1])[None, :] < lens[:, None]).long())
    e = (got - ref).abs().max().item()
    print(f"parity mode ITM logits vs fp32 oracle: max|d| {e:.2e} (plain f16 path: asserted 2e-3)")
    assert e < 2e-4
    # ---- a CapFiltEngine run in parity mode: captions and kept lists of the oracle pipeline
    cfg = dict(caption=True, filter=True, filter_generated_only=True, keep_original_caption=False, threshold=0.4,
               filter_mode="max_filter", generation_mode="beam", image_size=224, vit="base")
    eng = CapFiltEngine(cfg, DEV, captioner=cap, filterer=itm)
    Nv, Fv = 2, 4
    frames = synthetic_frames(Nv, Fv, first_video=11)
    items = [dict(video_id=f"video{v}", text=[]) for v in range(Nv)]
    eng.process(items, torch.from_numpy(frames).to(DEV))
    prompt = cap.prompt_ids(1, "cpu")[0].long().numpy()
    for v in range(Nv):
        xv = clip_ref.preprocess_u8(frames[v])
        caps_ref = pipeline_ref.caption_video(sd_cap, xv, prompt, tok, cap.prompt, dedup=True)
        assert eng.last_frame_captions[v * Fv:(v + 1) * Fv] == caps_ref, v
        kept, probs = pipeline_ref.filter_video(sd_itm, xv, items[v]["unfiltered_text"], tok, 0.4, return_probs=True, dedup=True)      # (same results as the per-caption ViT schedule: tests/test_oracle_cpu.py::test_deduplicated_cpu_schedule...)
        if all(abs(float(np.max(p)) - 0.4) > 1e-4 for p in probs):
            assert items[v]["text"] == kept, v


# ------------------------------------------------------------------------------- CLIP in the parity mode (round 4)
@pytest.fixture(scope="module")
def parity_clip():
    from vidil_amd.clip import CLIPModel
    from vidil_amd.packing import set_compute_dtype, set_parity_mode

    torch.manual_seed(0)
    clip = CLIPModel().eval()
    perturb_(clip, 102)
    sd = {k: v.clone() for k, v in clip.state_dict().items()}
    clip = clip.to(DEV)
    set_compute_dtype("f16", clip)
    set_parity_mode(True, clip)
    return clip, sd


def test_parity_mode_clip_embeddings_vs_fp32_oracle(parity_clip):
    """CLIP ViT-B/32 image embeddings and text embeddings (run_visual_tokenization.py:83-96,135-143: HF CLIPModel
    image_embeds / text_embeds, unit norm) with both towers on error-compensated operands, against the fp32 oracle
    (oracle/clip_ref.py, itself pinned to transformers' CLIPModel): 2e-5 / 6e-5 where the plain 16-bit towers are asserted at 5e-4."""
    from oracle import clip_ref

    clip, sd = parity_clip
    u8 = synthetic_frames(1, 8, first_video=5)[0]
    x = clip_ref.preprocess_u8(u8)
    with torch.no_grad():
        ref = clip_ref.image_embeds(sd, x)
    got = clip.encode_image_u8(torch.from_numpy(u8).to(DEV)).cpu()
    e_img = (got - ref).abs().max().item()
    got_f32 = clip.encode_image(x.to(DEV)).cpu()
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(1000, 40000, (6, 12), generator=g)
    ids[:, 0] = 49406
    for r, n in enumerate((5, 7, 9, 11, 12, 6)):
        ids[r, n - 1] = 49407
        ids[r, n:] = 0
    mask = (torch.arange(12)[None, :] < torch.tensor([5, 7, 9, 11, 12, 6])[:, None]).long()
    with torch.no_grad():
        tref = clip_ref.text_embeds(sd, ids, mask)
    tgot = clip.encode_text(ids.to(DEV), mask.to(DEV)).cpu()
    e_txt = (tgot - tref).abs().max().item()
    print(f"parity-mode CLIP ViT-B/32 vs fp32 oracle: image embeds max|d| {e_img:.2e} (f32 entry point {(got_f32 - ref).abs().max().item():.2e}), "
          f"text embeds {e_txt:.2e}  (plain 16-bit towers: asserted 5e-4)")
    # measured: image 3.1e-7, text 4.7e-7 (with the 16-bit attention kernels, $VIDIL_PARITY_ATTN=16: 5.9e-6 / 3.3e-5)
    assert e_img < 3e-6 and e_txt < 3e-6 and (got_f32 - ref).abs().max().item() < 3e-6
    assert (got.norm(dim=-1) - 1).abs().max().item() < 1e-5


def test_parity_mode_visual_token_indices_end_to_end_equal_the_reference_form(parity_clip):
    """BASELINE: "top-k visual-token indices bit-exact" — END TO END, frames -> CLIP tower -> ontology scan -> per-frame
    top-5 per category, against the reference FORM on the fp32 oracle's embeddings (`image_embeds @ text_embeds.t()` +
    `np.argsort(score)[::-1][:5]`, run_visual_tokenization.py:276,298-308), at config 1's shape per video (8 frames) and
    the vg ontology sizes (19,958 / 15,026 / 365 / 7,410 classes).  With the tower in the parity mode what separates the
    device's scores from the oracle's is the fp32 summation order, so a rank is compared exactly unless the ORACLE's own
    adjacent scores are closer than 5e-6 — and that mask is bounded at 5 % (round 3, 16-bit tower: gap 1.5e-3, 41 % masked)."""
    from oracle import clip_ref
    from test_models_gpu import _compare_visual_tokens_rank_by_rank, _ontology
    from vidil_amd.visual_tokenization import VisualTokenizer

    clip, sd = parity_clip
    Nv, F = 3, 8
    u8 = synthetic_frames(Nv, F, first_video=7)
    emb, texts = _ontology(sizes=dict(objects=19958, attributes=15026, scenes=365, verbs=7410))
    cfg = dict(topk_visualize=5)
    vt = VisualTokenizer(cfg, clip, texts, emb, DEV)
    toks = vt.process([f"video{v}" for v in range(Nv)], torch.from_numpy(u8).to(DEV), [[] for _ in range(Nv)])
    GAP = 5e-6          # (the device's embeddings are within 3e-7 of the oracle's with the mode's f32 attention; two fp32 matmuls of
    #                      512 terms in different summation orders still differ by ~1e-6 on cosine scores)
    ranks = masked = 0
    for v in range(Nv):
        with torch.no_grad():
            ie = clip_ref.image_embeds(sd, clip_ref.preprocess_u8(u8[v]))
        r, m = _compare_visual_tokens_rank_by_rank(toks[f"video{v}"], ie, emb, texts, F, GAP)
        ranks += r
        masked += m
    print(f"parity-mode e2e visual tokens: {ranks - masked}/{ranks} ranks compared exactly and equal "
          f"({masked} lie within {GAP} of a neighbour in the oracle's own scores)")
    assert masked <= 0.05 * ranks, (masked, ranks)


@pytest.mark.slow      # (round 4's mix, superseded by the qualified configuration: bench.py `secondary.parity_mix_full_step` only)
def test_parity_mix_plain_vit_with_compensated_decoder_keeps_caption_logits_within_1e_3(parity_captioner):
    """VERDICT r3 #2c — the CHEAPEST mix that still meets "caption logits within 1e-3" as an absolute bound
    (tests/probes/probe_parity_mix.py swept it: ViT blocks compensated k = 0 .. 12 -> worst pass 7.6e-4 .. 4.3e-4 at x1.34 ..
    x1.68 of the plain caption path): the ViT on PLAIN f16 operands (VisionTransformer.set_parity_last_blocks(0)), the cross
    K|V projection, the 12 decoder layers and the LM head on error-compensated operands.  All 16 teacher-forced passes."""
    from oracle import beam_ref, clip_ref, med_ref, vit_ref
    from vidil_amd.blip import DecoderSession

    from vidil_amd.packing import set_parity_attention

    cap, sd = parity_captioner
    cap.visual_encoder.set_parity_last_blocks(0)
    set_parity_attention("16", cap)            # the mix as bench.py times it: MFMA attention kernels (its error is the plain ViT's)
    cap.__dict__.pop("_decode_state", None)
    try:
        B, nb = 3, 3
        u8 = synthetic_frames(1, B)[0]
        with torch.no_grad():
            y_ref = vit_ref.vit_forward(sd, clip_ref.preprocess_u8(u8))
        enc3 = y_ref.repeat_interleave(nb, dim=0)
        state, otrace, calls = {}, [], []

        def step(ids, beam_idx):
            calls.append((ids.copy(), None if beam_idx is None else beam_idx.copy()))
            with torch.no_grad():
                past = None if beam_idx is None else med_ref.reorder_cache(state["cache"], torch.from_numpy(beam_idx))
                lg, state["cache"] = med_ref.decoder_logits(sd, torch.from_numpy(ids), enc3, past)
            return lg.numpy()

        prompt = cap.prompt_ids(B, "cpu").long().numpy()
        beam_ref.beam_search(step, prompt, num_beams=nb, max_length=20, min_length=5, eos_token_id=102, pad_token_id=0, trace=otrace)
        y32, y3 = cap.visual_encoder.forward_u8(torch.from_numpy(u8).to(DEV), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
        e_vit = (y32.cpu() - y_ref).abs().max().item()
        assert 5e-4 < e_vit < 1e-2, e_vit            # (the tower really ran on plain operands)
        sess = DecoderSession(cap.text_decoder, y3, B, nb, 20)
        worst = []
        for s_, (ids, beam_idx) in enumerate(calls[:len(otrace)]):
            if s_ == 0:
                lg = sess.prefill(torch.from_numpy(ids).to(torch.int32).reshape(-1).to(DEV), ids.shape[1])
            else:
                lg = sess.step(torch.from_numpy(ids[:, -1].copy()).to(torch.int32).to(DEV),
                               torch.from_numpy(beam_idx).to(torch.int32).to(DEV), ids.shape[1] - 1)
            worst.append((lg.cpu() - torch.from_numpy(otrace[s_]["logits"])).abs().max().item())
        print(f"parity MIX (plain ViT, compensated decoder + head + cross K|V), 16 passes: worst {max(worst):.2e} absolute "
              f"(asserted <= {ABS_TOL:g}); ViT output error {e_vit:.1e}")
        assert len(worst) == 16 and max(worst) <= ABS_TOL, worst
    finally:
        cap.visual_encoder.set_parity_last_blocks(None)
        set_parity_attention(None, cap)
        cap.__dict__.pop("_decode_state", None)


# ------------------------------------------------------------------------------- vidil_attention_f32 (round 4; arith 1: round 5)
@pytest.mark.parametrize("arith", [0, 1])    # 0: f32 arithmetic (f32-input MFMA / VALU); 1: split-operand 16-bit MFMA
@pytest.mark.parametrize("Bq,H,Nq,Nk,kv_group,causal,use_len", [
    (3, 12, 197, 197, 1, False, False),     # ViT self-attention, read in place from a [M, 3C] projection output
    (4, 8, 13, 13, 1, True, True),          # CLIP text tower: causal + kv_len
    (6, 12, 3, 197, 3, False, False),       # decode cross-attention: 3 beams share an image's K | V
    (5, 4, 35, 70, 1, False, False),        # two key chunks
    (6, 12, 1, 197, 3, False, False),       # 3 rows per unit: the VALU kernel (idle row groups skipped)
    (2, 4, 300, 300, 1, False, False),      # more than 128 rows per unit: several workgroups of the f32-MFMA kernel
    (3, 4, 40, 577, 1, True, False),        # causal with more keys than rows (19 key tiles)
])
def test_attention_f32_vs_float64(Bq, H, Nq, Nk, kv_group, causal, use_len, arith):
    k = _k()
    C = H * 64
    Bk = Bq // kv_group
    self_attn = Nq == Nk and kv_group == 1
    if self_attn:
        qkv = _rand(Bq * Nq, 3 * C, seed=60).to(DEV)
        q, kk, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    else:
        q = _rand(Bq * Nq, C, seed=61).to(DEV)
        kv = _rand(Bk, Nk, 2 * C, seed=62).to(DEV)
        kk, v = kv[..., :C], kv[..., C:]
    kv_len = None
    if use_len:
        kv_len = torch.tensor([Nk, 5, 9, 1][:Bq], dtype=torch.int32)
    out32 = torch.zeros(Bq * Nq, C, dtype=torch.float32, device=DEV)
    out3 = torch.zeros(Bq * Nq, 3 * C, dtype=torch.float16, device=DEV)
    args = dict(Bq=Bq, H=H, Nq=Nq, Nk=Nk, kv_group=kv_group, causal=causal, kv_len=None if kv_len is None else kv_len.to(DEV), arith=arith)
    k.attention_f32(q, kk, v, out32, **args)
    k.attention_f32(q, kk, v, out3, **args)
    qd = q.double().cpu().view(Bq, Nq, H, 64).permute(0, 2, 1, 3)
    kd = kk.double().cpu().reshape(Bk, Nk, H, 64).permute(0, 2, 1, 3).repeat_interleave(kv_group, 0)
    vd = v.double().cpu().reshape(Bk, Nk, H, 64).permute(0, 2, 1, 3).repeat_interleave(kv_group, 0)
    s = (qd @ kd.transpose(-1, -2)) * 0.125
    keys = torch.arange(Nk)
    if causal:
        s = s.masked_fill(keys[None, :] > torch.arange(Nq)[:, None], float("-inf"))
    if kv_len is not None:
        s = s.masked_fill(keys[None, None, None, :] >= kv_len.long()[:, None, None, None], float("-inf"))
    ref = (torch.softmax(s, -1) @ vd).permute(0, 2, 1, 3).reshape(Bq * Nq, C)
    e32 = (out32.cpu().double() - ref).abs().max().item()
    e3 = (_join(out3.cpu()).double() - ref).abs().max().item()
    print(f"attention_f32 arith={arith} {Bq}x{H}x{Nq}x{Nk}: max|d| vs float64 {e32:.2e} (f32 rows) {e3:.2e} ([hi | lo | hi] rows)")
    assert e32 < 3e-6 and e3 < 5e-6, (e32, e3)        # (the 16-bit MFMA kernels: ~4e-4 on the same data)


@pytest.mark.parametrize("B,nb,Nq,Nk", [(5, 3, 1, 197), (4, 3, 4, 197), (3, 1, 4, 50), (2, 3, 1, 577)])
def test_attention_f32_split_form_on_16bit_fragment_tiles(B, nb, Nq, Nk):
    """arith 1 + kv16 (the decode steps' cross-attention of the parity mode, round 5): f32 queries read in place and split into
    hi + lo, the probabilities split alike, K / V = the 16-bit fragment tiles of the plain path.  Against float64 attention on
    exactly those stored K / V values: only the split arithmetic is left (~1e-6); against the UNROUNDED K / V the difference is
    the 16-bit rounding of K / V that tests/probes/probe_precision_design.py prices."""
    k = _k()
    from vidil_amd.kernels import kv_tile_offsets
    H = 12
    C = H * 64
    Tc = (Nk + 31) // 32 * 32
    q = _rand(B * nb * Nq, 3 * C, seed=80).to(DEV)[:, C:2 * C]            # (a column slice: row stride 3C)
    kf, vf = _rand(B, H, Nk, 64, seed=81), _rand(B, H, Nk, 64, seed=82)
    k16, v16 = kf.half(), vf.half()
    ko, vo = kv_tile_offsets(Nk)
    kt = torch.zeros(B, H, Tc * 64, dtype=torch.float16)
    vt = torch.zeros(B, H, Tc * 64, dtype=torch.float16)
    kt[:, :, ko.reshape(-1)] = k16.reshape(B, H, -1)
    vt[:, :, vo.reshape(-1)] = v16.reshape(B, H, -1)
    out3 = torch.zeros(B * nb * Nq, 3 * C, dtype=torch.float16, device=DEV)
    k.attention_f32(q, kt.view(B, H, Tc, 64).to(DEV), vt.view(B, H, Tc, 64).to(DEV), out3, Bq=B * nb, H=H, Nq=Nq, Nk=Nk, kv_rows=Tc,
                    kv_group=nb, arith=1, kv16=True)
    qd = q.double().cpu().view(B * nb, Nq, H, 64).permute(0, 2, 1, 3)
    kd = k16.double().repeat_interleave(nb, 0)
    vd = v16.double().repeat_interleave(nb, 0)
    ref = (torch.softmax((qd @ kd.transpose(-1, -2)) * 0.125, -1) @ vd).permute(0, 2, 1, 3).reshape(B * nb * Nq, C)
    e = (_join(out3.cpu()).double() - ref).abs().max().item()
    print(f"attention_f32 split / kv16 {B}x{nb}x{Nq}x{Nk}: max|d| vs float64 on the stored K / V {e:.2e}")
    assert e < 5e-6, e


def test_attention_f32_arena_form_follows_the_ancestry_table():
    k = _k()
    R, H, Tcap, n_keys = 12, 12, 20, 7
    C = H * 64
    g = torch.Generator().manual_seed(70)
    arena_k = _rand(Tcap, R, C, seed=71).to(DEV)
    arena_v = _rand(Tcap, R, C, seed=72).to(DEV)
    anc = torch.randint(0, R, (R, Tcap), generator=g, dtype=torch.int32)
    q = _rand(R, 3 * C, seed=73).to(DEV)[:, :C]
    out = torch.zeros(R, C, dtype=torch.float32, device=DEV)
    k.attention_f32(q, arena_k, arena_v, out, Bq=R, H=H, Nq=1, Nk=n_keys, anc=anc.to(DEV), arena_rows=R)
    kk = torch.stack([arena_k.cpu()[torch.arange(n_keys), anc[r, :n_keys].long()] for r in range(R)]).double()   # [R, n_keys, C]
    vv = torch.stack([arena_v.cpu()[torch.arange(n_keys), anc[r, :n_keys].long()] for r in range(R)]).double()
    qd = q.cpu().double().view(R, H, 1, 64)
    s = (qd @ kk.view(R, n_keys, H, 64).permute(0, 2, 3, 1)) * 0.125
    ref = (torch.softmax(s, -1) @ vv.view(R, n_keys, H, 64).permute(0, 2, 1, 3)).reshape(R, C)
    assert (out.cpu().double() - ref).abs().max().item() < 3e-6


# ------------------------------------------------------------------------------- out16_split3 (round 5)
@pytest.mark.parametrize("M,N,K,act", [(300, 192, 384, 1), pytest.param(4096, 3072, 2304, 1, marks=pytest.mark.slow), pytest.param(99000, 768, 1536, 0, marks=pytest.mark.slow),
                                           (33000, 3072, 1536, 2), pytest.param(10752, 3072, 2304, 1, marks=pytest.mark.slow)])
def test_gemm_f32_epilogue_writes_split3_operand_rows_itself(M, N, K, act):
    """vidil_gemm_args.out16_split3: the f32 epilogue (bias + activation in f32) hands its result over as [hi | lo | hi] rows —
    bit for bit what vidil_split3_f32 makes of the same GEMM's f32 output, on the 8-wave kernel (small grids) and the 4-wave one;
    with `out` given as well the f32 rows are written too and equal the plain launch's."""
    k = _k()
    a = (_rand(M, K, seed=90) * 0.5).half().to(DEV)
    w = (_rand(N, K, seed=91) * 0.05).half().to(DEV)
    b = _rand(N, seed=92).to(DEV)
    ref32 = k.gemm(a, w, b, out=torch.empty(M, N, dtype=torch.float32, device=DEV), act=act)
    ref3 = k.split3(ref32, torch.empty(M, 3 * N, dtype=torch.float16, device=DEV))
    got3 = torch.zeros(M, 3 * N, dtype=torch.float16, device=DEV)
    k.gemm(a, w, b, split3_out=got3, act=act)
    assert torch.equal(got3, ref3), (got3.float() - ref3.float()).abs().max().item()
    got3b = torch.zeros(M, 3 * N, dtype=torch.float16, device=DEV)
    out32 = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    k.gemm(a, w, b, split3_out=got3b, out=out32, act=act)
    assert torch.equal(got3b, ref3) and torch.equal(out32, ref32)
    # split3_planes=2 (the consumer is a split_k launch in the K-loop form, which reads planes hi | lo only): the third plane is
    # left untouched
    got2 = torch.full((M, 3 * N), 7.0, dtype=torch.float16, device=DEV)
    k.gemm(a, w, b, split3_out=got2, act=act, split3_planes=2)
    assert torch.equal(got2[:, :2 * N], ref3[:, :2 * N]) and bool((got2[:, 2 * N:] == 7.0).all())
    print(f"split3_out {M}x{N}x{K} act {act}: {k.gemm_kernel_name(a, w, b, split3_out=got3, act=act)}")


# ------------------------------------------------------------------------------- split_k: the in-loop compensated product (round 5)
@pytest.mark.parametrize("M,N,K,act", [(197 * 3, 2304, 768, 0), (12, 30524, 768, 0), (197 * 130, 768, 3072, 0), (70000, 3072, 768, 1),
                                       (45000, 1024, 640, 2), (300, 768, 192, 0)])
def test_split_k_gemm_reproduces_the_fp32_product_at_every_size_with_the_same_bits(M, N, K, act):
    """vidil_gemm_args.split_k: the three products x_hi.W_hi + x_hi.W_lo + x_lo.W_hi formed inside ONE K loop over Kl (gemm4w's C3
    form: 128- or 256-row tiles by grid size) — as close to the fp64 product as the K-tripled launch, and BIT-IDENTICAL for a row
    whether it is multiplied alone (a few rows: 128-row tiles) or inside the full problem (256-row tiles)."""
    from vidil_amd.packing import w3

    k = _k()
    x = _rand(M, K, seed=40)
    w = _rand(N, K, scale=0.03, seed=41)
    bias = _rand(N, seed=42)
    pre = x.double() @ w.double().t() + bias.double()
    ref = pre if act == 0 else (torch.nn.functional.gelu(pre) if act == 1 else pre * torch.sigmoid(1.702 * pre))
    a3 = k.split3(x.to(DEV), torch.empty(M, 3 * K, dtype=torch.float16, device=DEV))
    w3d = w3(w, dtype=torch.float16).to(DEV)
    name = k.gemm_kernel_name(a3, w3d, bias.to(DEV), out_dtype=torch.float32, act=act, split_k=True)
    assert name.startswith("gemm4w_kernel") and name.endswith("true>"), name
    got = k.gemm(a3, w3d, bias.to(DEV), out_dtype=torch.float32, act=act, split_k=True)
    tripled = k.gemm(a3, w3d, bias.to(DEV), out_dtype=torch.float32, act=act)          # the K-tripled form of rounds 3-4
    e, e3 = (got.cpu().double() - ref).abs().max().item(), (tripled.cpu().double() - ref).abs().max().item()
    print(f"M={M} N={N} K={K} act={act}: {name}: max|d| vs fp64 {e:.2e} (K-tripled launch {e3:.2e})")
    assert e < 1e-5 * max(1.0, ref.abs().max().item()) + 1e-5 and e < 3 * e3 + 1e-6
    # a handful of rows alone: another tile height, another grid — the same bits
    rows = torch.tensor([0, 1, M // 2, M - 1][: min(4, M)])
    sub = k.gemm(a3[rows.to(DEV)].contiguous(), w3d, bias.to(DEV), out_dtype=torch.float32, act=act, split_k=True)
    assert torch.equal(sub, got[rows.to(DEV)])
    # the [hi | lo | hi] hand-over and the residual form on the same path
    res = _rand(M, N, seed=43).to(DEV)
    got_r = k.gemm(a3, w3d, bias.to(DEV), out=res.clone(), resid=res, act=act, split_k=True)
    assert torch.equal(got_r, got + res)
    if N % 8 == 0:
        s3 = k.gemm(a3, w3d, bias.to(DEV), split3_out=torch.zeros(M, 3 * N, dtype=torch.float16, device=DEV), act=act, split_k=True)
        assert torch.equal(s3, k.split3(got, torch.empty(M, 3 * N, dtype=torch.float16, device=DEV)))


def test_split_k_gemm_per_head_scatter_and_patch_epilogues_equal_the_k_tripled_launch_to_rounding():
    """EPI_HEADS (the cross K | V fragment tiles, T >= 8) and EPI_PATCH on the in-loop compensated product: 16-bit / f32 outputs
    within one rounding of the K-tripled launch's (the f32 sums are taken in another order)."""
    from vidil_amd.packing import w3

    k = _k()
    B, T, H, C = 40, 197, 12, 768
    x = _rand(B * T, C, seed=50)
    w = _rand(2 * C, C, scale=0.03, seed=51)
    bias = _rand(2 * C, seed=52).to(DEV)
    a3 = k.split3(x.to(DEV), torch.empty(B * T, 3 * C, dtype=torch.float16, device=DEV))
    w3d = w3(w, dtype=torch.float16).to(DEV)
    Tc = 224
    outs = []
    for sk in (True, False):
        kk = torch.zeros(B, H, Tc, 64, dtype=torch.float16, device=DEV)
        vv = torch.zeros(B, H, Tc, 64, dtype=torch.float16, device=DEV)
        k.gemm(a3, w3d, bias, heads=dict(k=kk, vt=vv, T=T, H=H, part0=1, t_off=0, Tk_cap=Tc, tiled=True), split_k=sk)
        outs.append((kk, vv))
    for a_, b_ in zip(outs[0], outs[1]):
        # one unit in the last place of the 16-bit value where the two f32 sums straddle a rounding boundary (~0.1 % of the
        # elements): compared as bit patterns — same sign, ordinals at most 1 apart
        ia, ib = a_.view(torch.int16).int(), b_.view(torch.int16).int()
        oa, ob = torch.where(ia >= 0, ia, -(ia & 0x7FFF)), torch.where(ib >= 0, ib, -(ib & 0x7FFF))      # monotonic ordinals (+-0 -> 0)
        ulps = (oa - ob).abs()
        # (small values: the two f32 sums differ by ~1e-6 ABSOLUTE — sums of O(1) terms in another order — which is many ulps of a
        #  16-bit value near zero; one ulp OR that absolute difference)
        ok = (ulps <= 1) | ((a_.float() - b_.float()).abs() <= 4e-6)
        frac = (ulps > 0).float().mean().item()
        assert bool(ok.all()) and frac < 0.02, ((a_.float() - b_.float()).abs()[~ok].max().item() if not bool(ok.all()) else 0.0, frac)
    # patch embedding: [B*P, 3*768] split rows -> f32 stream rows (m + m / tpi + 1)
    P = 196
    xp = _rand(3 * P, 768, seed=53)
    wp = _rand(C, 768, scale=0.03, seed=54)
    pos = _rand(P + 1, C, seed=55).to(DEV)
    ap = k.split3(xp.to(DEV), torch.empty(3 * P, 3 * 768, dtype=torch.float16, device=DEV))
    wpd = w3(wp, dtype=torch.float16).to(DEV)
    res = []
    for sk in (True, False):
        out = torch.zeros(3 * (P + 1), C, dtype=torch.float32, device=DEV)
        k.gemm(ap, wpd, bias[:C].contiguous(), patch=dict(out=out, pos=pos, tpi=P), split_k=sk)
        res.append(out)
    assert (res[0] - res[1]).abs().max().item() < 2e-5


def test_two_plane_split_rows_of_layernorm_and_attention_f32_leave_the_third_plane_untouched():
    """VIDIL_DT_SPLIT2 / out_mode 3 (round 5): producers whose consumer is a split_k GEMM in the K-loop form write planes hi | lo
    only — bit-identical to the first two planes of the three-plane output, third plane not touched."""
    k = _k()
    M, D = 777, 768
    x = _rand(M, D, seed=120).to(DEV)
    g, b = (1 + 0.1 * _rand(D, seed=121)).to(DEV), _rand(D, seed=122).to(DEV)
    o3 = torch.zeros(M, 3 * D, dtype=torch.float16, device=DEV)
    o2 = torch.full((M, 3 * D), 7.0, dtype=torch.float16, device=DEV)
    k.layernorm(x, g, b, 1e-6, out16=o3, split3=True)
    k.layernorm(x, g, b, 1e-6, out16=o2, split3=True, planes=2)
    assert torch.equal(o2[:, :2 * D], o3[:, :2 * D]) and bool((o2[:, 2 * D:] == 7.0).all())
    H, C = 12, 768
    for (Bq, Nq, Nk, kvg, arith) in [(3, 197, 197, 1, 1), (6, 1, 197, 3, 1), (4, 40, 40, 1, 0)]:
        if Nq == Nk and kvg == 1:
            qkv = _rand(Bq * Nq, 3 * C, seed=123).to(DEV)
            q, kk, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        else:
            q = _rand(Bq * Nq, C, seed=124).to(DEV)
            kv = _rand(Bq // kvg, Nk, 2 * C, seed=125).to(DEV)
            kk, v = kv[..., :C], kv[..., C:]
        a3 = torch.zeros(Bq * Nq, 3 * C, dtype=torch.float16, device=DEV)
        a2 = torch.full((Bq * Nq, 3 * C), 7.0, dtype=torch.float16, device=DEV)
        k.attention_f32(q, kk, v, a3, Bq=Bq, H=H, Nq=Nq, Nk=Nk, kv_group=kvg, arith=arith)
        k.attention_f32(q, kk, v, a2, Bq=Bq, H=H, Nq=Nq, Nk=Nk, kv_group=kvg, arith=arith, planes=2)
        assert torch.equal(a2[:, :2 * C], a3[:, :2 * C]) and bool((a2[:, 2 * C:] == 7.0).all()), (Bq, Nq, Nk)


def test_parity_qualified_results_do_not_depend_on_batch_composition():
    """The property behind 1/2/4/8-GPU equality, in the PARITY-QUALIFIED configuration (captioner + CLIP compensated, filter plain):
    a video processed alone (a few hundred GEMM rows: 128-row tiles of the K-loop kernel, 4-wave attention workgroups) gives the
    bits it gives inside a batch of 36 videos (256-row tiles) — ViT output and CLIP embeddings compared bit for bit, captions, kept
    lists and visual tokens as strings."""
    from vidil_amd.blip import BLIP_Decoder
    from vidil_amd.blip_itm import BLIP_ITM
    from vidil_amd.capfilt import CapFiltEngine
    from vidil_amd.clip import CLIPModel
    from vidil_amd.packing import set_compute_dtype, set_parity_mode
    from vidil_amd.tokenizer import SyntheticBertTokenizer
    from vidil_amd.visual_tokenization import CATEGORIES, VisualTokenizer
    from oracle import clip_ref

    torch.manual_seed(0)
    tok = SyntheticBertTokenizer()
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=tok).eval()
    itm = BLIP_ITM(image_size=224, vit="base", tokenizer=tok).eval()
    clip = CLIPModel().eval()
    for i, m in enumerate((cap, itm, clip)):
        perturb_(m, 100 + i)
    cap, itm, clip = cap.to(DEV), itm.to(DEV), clip.to(DEV)
    set_compute_dtype("f16", cap, clip)
    set_compute_dtype("bf16", itm)
    set_parity_mode(True, cap, clip)
    Nv, F = 36, 8
    u8 = torch.from_numpy(synthetic_frames(Nv, F, first_video=300)).to(DEV)
    cfg = dict(caption=True, filter=True, filter_generated_only=True, keep_original_caption=False, threshold=0.4,
               filter_mode="max_filter", generation_mode="beam", image_size=224, vit="base", topk_visualize=5)
    eng = CapFiltEngine(cfg, DEV, captioner=cap, filterer=itm)
    g = torch.Generator().manual_seed(5)
    emb, texts = {}, {}
    for key, n in zip(CATEGORIES, (500, 300, 60, 200)):
        e = torch.randn(n, 512, generator=g)
        emb[key], texts[key] = e / e.norm(dim=-1, keepdim=True), [f"{key}_{i}" for i in range(n)]
    vt = VisualTokenizer(cfg, clip, texts, emb, DEV)

    def run(lo, hi):
        items = [dict(video_id=f"video{v}", text=[]) for v in range(lo, hi)]
        eng.process(items, u8[lo:hi])
        t = vt.process([it["video_id"] for it in items], u8[lo:hi], [[] for _ in items])
        y32, _ = cap.visual_encoder.forward_u8(u8[lo:hi].reshape(-1, 224, 224, 3), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
        ce = clip.encode_image_u8(u8[lo:hi].reshape(-1, 224, 224, 3))
        return items, t, y32.clone(), ce.clone()

    all_items, all_t, all_y, all_c = run(0, Nv)
    for v in (0, 17, 35):
        it, t, y, c = run(v, v + 1)
        assert torch.equal(y, all_y[v * F:(v + 1) * F]), "ViT output bits depend on the batch"
        assert torch.equal(c, all_c[v * F:(v + 1) * F]), "CLIP embedding bits depend on the batch"
        assert it[0]["unfiltered_text"] == all_items[v]["unfiltered_text"] and it[0]["text"] == all_items[v]["text"]
        assert t[f"video{v}"] == all_t[f"video{v}"]


def test_split_k_two_plane_operands_are_refused_where_the_plain_product_would_read_the_third_plane():
    """producers leave plane 2 of their [hi | lo | hi] rows unwritten only where the CONSUMER call takes the
    K-loop form — asked per call (vidil_gemm_split_k_serves) — and a consumer told its A rows hold two planes (split_k = 2) fails
    with EINVAL instead of running the plain K = 3 Kl product over the unwritten plane.  The per-head epilogue with fewer than 8
    tokens per sequence is such a call (tiny image configs: Te < 8)."""
    from vidil_amd import kernels as K
    from vidil_amd.packing import w3

    torch.manual_seed(3)
    C, H, T, B = 128, 2, 5, 6
    M = B * T
    x = torch.randn(M, C, device=DEV)
    w = torch.randn(2 * C, C) * 0.05
    a3 = K.split3(x, torch.empty((M, 3 * C), dtype=torch.float16, device=DEV))
    w3_ = w3(w, dtype=torch.float16).to(DEV)
    bias = torch.zeros(2 * C, device=DEV)
    k = torch.zeros((B, H, T, 64), dtype=torch.float16, device=DEV)
    v = torch.zeros((B, H, T, 64), dtype=torch.float16, device=DEV)
    heads = dict(k=k, vt=v, T=T, H=H, part0=1, t_off=0, Tk_cap=T, NP=0)
    out32 = torch.empty((M, 2 * C), dtype=torch.float32, device=DEV)
    assert K.split_k_in_loop()
    assert K.split_k_serves(a3, w3_, bias, out=out32)                  # f32 epilogue: the K-loop form at every size
    assert not K.split_k_serves(a3, w3_, bias, heads=heads)            # 5 tokens per sequence: the plain K = 3 Kl product
    # three valid planes: served by the plain product, same values as the f32 epilogue's
    K.gemm(a3, w3_, bias, heads=heads, split_k=True)
    K.gemm(a3, w3_, bias, out=out32, split_k=True, a_planes=2)
    torch.cuda.synchronize()
    ref = out32.view(B, T, 2, H, 64)
    assert torch.allclose(k.float(), ref[:, :, 0].permute(0, 2, 1, 3), atol=2e-3, rtol=2e-3)
    # two planes only (NaN in the third, as a two-plane producer may leave it): refused, loudly
    a2 = a3.clone()
    a2[:, 2 * C:] = float("nan")
    with pytest.raises(K.VidilHipError, match="split_k=2"):
        K.gemm(a2, w3_, bias, heads=heads, split_k=True, a_planes=2)
    K.gemm(a2, w3_, bias, out=out32, split_k=True, a_planes=2)         # ... and the K-loop form never reads it
    torch.cuda.synchronize()
    assert torch.isfinite(out32).all()


def test_tiny_image_parity_captioner_has_no_unwritten_plane_on_its_path(monkeypatch):
    """The concrete case of ADVICE r5: a 32 x 32 image config has Te = 5 image tokens, so the cross K | V projection's per-head
    epilogue runs the plain product over ALL planes of the ViT's output rows — which the final LayerNorm therefore writes in full.
    With $VIDIL_POISON_SPLIT3 (NaNs in every plane a producer is allowed to skip) the logits stay finite and equal the unpoisoned run."""
    from vidil_amd.blip import BLIP_Decoder, DecoderSession
    from vidil_amd.packing import set_compute_dtype, set_parity_mode
    from vidil_amd.tokenizer import SyntheticBertTokenizer

    torch.manual_seed(0)
    cap = BLIP_Decoder(image_size=32, vit="base", tokenizer=SyntheticBertTokenizer()).eval().to(DEV)
    set_compute_dtype("f16", cap)
    set_parity_mode(True, cap)
    img = torch.randn(3, 3, 32, 32, device=DEV)
    prompt = cap.prompt_ids(3, DEV)
    P = prompt.shape[1]

    def logits():
        _, y3 = cap.visual_encoder.forward_both(img)
        sess = DecoderSession(cap.text_decoder, y3, 3, 3, 12, tiled_cross=True)
        return sess.prefill(prompt.contiguous().view(-1), P, shared=True).float().cpu()

    a = logits()
    monkeypatch.setenv("VIDIL_POISON_SPLIT3", "1")
    b = logits()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert torch.equal(a, b)


def test_split_attention_session_refuses_more_than_32_query_rows_per_image_with_a_clear_message():
    """ADVICE r5 (attention.hip kv16 form: at most 32 query rows per image): a decoder session on 16-bit cross K / V tiles whose
    prompt pass runs every beam row (P tokens x nb beams > 32 rows per image) is refused by the host with the remedy in the message
    instead of an EINVAL from the launch; the shared prompt pass (P rows per image) and tiled_cross=False both work."""
    from vidil_amd import kernels as K
    from vidil_amd.blip import BLIP_Decoder, DecoderSession
    from vidil_amd.packing import set_compute_dtype, set_parity_mode
    from vidil_amd.tokenizer import SyntheticBertTokenizer

    torch.manual_seed(0)
    cap = BLIP_Decoder(image_size=32, vit="base", tokenizer=SyntheticBertTokenizer()).eval().to(DEV)
    set_compute_dtype("f16", cap)
    set_parity_mode(True, cap)
    B, nb, P = 2, 3, 12                                    # 36 query rows per image in an unshared prompt pass
    _, y3 = cap.visual_encoder.forward_both(torch.randn(B, 3, 32, 32, device=DEV))
    ids = torch.randint(1000, 2000, (B * nb, P), dtype=torch.int32, device=DEV)
    sess = DecoderSession(cap.text_decoder, y3, B, nb, 20, tiled_cross=True)
    with pytest.raises(K.VidilHipError, match="tiled_cross=False"):
        sess.prefill(ids.reshape(-1), P)
    lg_shared = DecoderSession(cap.text_decoder, y3, B, nb, 20, tiled_cross=True).prefill(ids[::nb].reshape(-1).contiguous(), P, shared=True)
    lg_rows = DecoderSession(cap.text_decoder, y3, B, nb, 20, tiled_cross=False).prefill(ids[::nb].repeat_interleave(nb, 0).reshape(-1).contiguous(), P)
    torch.cuda.synchronize()
    assert torch.isfinite(lg_shared).all() and (lg_rows[::nb] - lg_shared).abs().max().item() < 1e-3


def test_clip_last_layer_on_class_token_rows_gives_the_full_layers_embeddings(parity_clip, monkeypatch):
    """Round 6: in the parity precision mode the CLIP vision tower runs its LAST layer for the class-token rows only (K | V for
    every token; query, attention output, out-proj, LayerNorm 2 and the MLP for one row per image — the pooled output reads nothing
    else, HF CLIPVisionTransformer.forward: `pooled_output = last_hidden_state[:, 0, :]`).  Same function: embeddings equal the
    full-layer run's to f32 rounding (the class token's attention runs in plain f32 arithmetic instead of the split-operand form)
    and stay within the mode's bound of the fp32 oracle; a frame's embedding does not depend on the batch around it."""
    from oracle import clip_ref

    clip, sd = parity_clip
    u8 = torch.from_numpy(synthetic_frames(1, 6, first_video=77)[0]).to(DEV)
    clip.cls_only_last_layer = False
    full = clip.encode_image_u8(u8).float().cpu()
    clip.cls_only_last_layer = True
    fast = clip.encode_image_u8(u8).float().cpu()
    alone = clip.encode_image_u8(u8[2:3].contiguous()).float().cpu()
    with torch.no_grad():
        ref = clip_ref.image_embeds(sd, clip_ref.preprocess_u8(u8.cpu().numpy()))
    d_ff, d_ref = (fast - full).abs().max().item(), (fast - ref).abs().max().item()
    print(f"\nCLS-only last layer vs full layer: max |d| {d_ff:.2e}; vs fp32 oracle {d_ref:.2e} (full layer: {(full - ref).abs().max().item():.2e})")
    assert d_ff < 2e-6 and d_ref < 3e-6
    assert torch.equal(alone[0], fast[2])
