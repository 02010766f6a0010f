"""Parity at a TRAINED-MODEL scale, without the weights (reference: models/med.py:501-545 LM head,
models/blip.py:332-354 checkpoint loading).  Every other float-parity test of this tree runs on random-init weights:
max|logit| ~ 2.5, LayerNorm gains ~ 1, residual rows with zero mean.  Here the same comparisons run on a synthetic
"trained-like" state dict (tests/common.py trained_like_): LM-head weights scaled so that max|logit| is 15-20, LayerNorm gains
with 10-50x outlier channels, residual rows ~1 sigma off zero, a [SEP] bias so that searches end at different lengths.

What is asserted, and the SCALE LAW that comes out of it (DESIGN.md §4 quotes the printed table):
  * parity precision mode: the error is PROPORTIONAL to the logit scale.  With the 16-bit MFMA attention kernels of round 3
    ($VIDIL_PARITY_ATTN=16) it was 2.3e-4 .. 2.5e-4 of max|logit| — "within 1e-3" ABSOLUTE held up to max|logit| ~ 4 and not above
    (1.9e-3 at scale 8, 3.7e-3 at 15.7) — because every GEMM operand was carried to 2^-21 but Q / K / V were still rounded to 16
    bits inside the attention kernels (tests/probes/probe_attention_rounding.py reproduces exactly that residual on the CPU).  With
    the mode's f32 attention (vidil_attention_f32, the default since round 4) it is 8e-6 .. 9e-6 of the scale: 1.4e-4 at
    max|logit| = 15.7.  The test asserts `<= PARITY_REL * scale` AND the absolute 1e-3 at every scale;
  * plain f16: 1.5e-3 .. 1.6e-3 of the logit scale on these statistics (outlier gains amplify the operand rounding: random-init
    weights give 0.85e-3), asserted at PLAIN_REL;
  * LayerNorm-folded tower vs the unfolded kernels on the same weights: within the fold's budget;
  * free-running captions (with searches ending at different lengths, i.e. through decode compaction) = the oracle's."""
import json
import os

import numpy as np
import pytest
import torch

from common import ROOT, synthetic_frames, trained_like_

pytestmark = pytest.mark.gpu
DEV = "cuda"
PARITY_REL = 3e-5         # parity mode, f32 attention: max|d logit| <= PARITY_REL * max|logit|   (measured 8.3e-6 .. 9.0e-6 at scales 2 / 8 / 16)
SPLIT_REL = 5e-5          # parity mode, split-operand attention in the PRODUCT's form (the default since round 5: the decode steps'
#                           cross-attention on the plain path's 16-bit K / V fragment tiles, Q and P split): the CPU probe
#                           (tests/probes/probe_precision_design.py) prices the 16-bit K / V at 3.0e-5 of the scale; 1e-3 absolute is asserted beside it
PLAIN_REL = 2.2e-3        # plain f16 operands (measured 1.5e-3 .. 1.6e-3 of the scale; random-init statistics: 0.85e-3)


def _build(head_scale, sep_bias=None):
    from vidil_amd.blip import BLIP_Decoder
    from vidil_amd.tokenizer import SyntheticBertTokenizer

    torch.manual_seed(0)
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=SyntheticBertTokenizer()).eval()
    info = trained_like_(cap, 300, head_scale=head_scale, sep_bias=sep_bias, stream_shift=8.0)
    sd = {k: v.clone() for k, v in cap.state_dict().items()}
    return cap, sd, info


def _teacher_forced(cap, sd, u8, nb=3, max_length=12, tiled_cross=False):
    """The oracle's beam search (its logits, its decisions) and the device's logits on the same decisions."""
    from oracle import beam_ref, clip_ref, med_ref, vit_ref
    from vidil_amd.blip import DecoderSession

    B = u8.shape[0]
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
    seqs, _ = beam_ref.beam_search(step, prompt, num_beams=nb, max_length=max_length, min_length=5, eos_token_id=102, pad_token_id=0,
                                   trace=otrace)
    y32, yop = cap.visual_encoder.forward_u8(torch.from_numpy(u8).to(DEV), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
    sess = DecoderSession(cap.text_decoder, yop, B, nb, max_length, tiled_cross=tiled_cross)
    rows = []
    for s, (ids, beam_idx) in enumerate(calls[:len(otrace)]):
        if s == 0:
            lg = sess.prefill(torch.from_numpy(ids).to(torch.int32).reshape(-1).to(DEV), ids.shape[1])
        else:
            lg = sess.step(torch.from_numpy(ids[:, -1].copy()).to(torch.int32).to(DEV),
                           torch.from_numpy(beam_idx).to(torch.int32).to(DEV), ids.shape[1] - 1)
        ref = torch.from_numpy(otrace[s]["logits"])
        d = (lg.cpu() - ref).abs()
        rows.append((d.max().item(), ref.abs().max().item()))
    e_vit = (y32.cpu() - y_ref).abs().max().item()
    return rows, seqs, e_vit, yop, y_ref


@pytest.mark.parametrize("head_scale", [pytest.param(0.25, marks=pytest.mark.slow), pytest.param(1.0, marks=pytest.mark.slow), 2.0])      # max|logit| ~ 3.5 / 10 / 19 (the transform LayerNorm's outlier gains carry most of it)
def test_caption_logits_at_trained_like_statistics_plain_and_parity(head_scale):
    from vidil_amd.packing import set_compute_dtype, set_parity_mode

    cap, sd, info = _build(head_scale)
    cap = cap.to(DEV)
    set_compute_dtype("f16", cap)
    u8 = synthetic_frames(1, 2, first_video=21)[0]
    plain, _, e_vit_plain, _, y_ref = _teacher_forced(cap, sd, u8)
    from vidil_amd.packing import set_parity_attention
    set_parity_mode(True, cap)
    set_parity_attention("f32", cap)
    par, _, e_vit_par, _, _ = _teacher_forced(cap, sd, u8)
    # the split-operand attention (round 5, the mode's default), as the product runs it: image K / V of the decode steps as
    # 16-bit fragment tiles (generate_ids' tiled_cross) — and once more with f32 K / V rows (every operand split)
    set_parity_attention("split", cap)
    spl, _, e_vit_spl, _, _ = _teacher_forced(cap, sd, u8, tiled_cross=True)
    spl_f32kv, _, _, _, _ = _teacher_forced(cap, sd, u8, tiled_cross=False)
    set_parity_attention(None, cap)
    set_parity_mode(False, cap)
    scale = max(s for _, s in plain)
    worst_plain, worst_par = max(e for e, _ in plain), max(e for e, _ in par)
    worst_spl, worst_spl_f32kv = max(e for e, _ in spl), max(e for e, _ in spl_f32kv)
    mu_sigma = float((y_ref.mean(dim=-1).abs() / y_ref.std(dim=-1)).mean())
    rec = dict(head_scale=head_scale, logit_scale=scale, passes=len(plain), plain_f16_max_abs=worst_plain, plain_rel=worst_plain / scale,
               parity_max_abs=worst_par, parity_rel=worst_par / scale, split_max_abs=worst_spl, split_rel=worst_spl / scale,
               split_f32kv_max_abs=worst_spl_f32kv, split_f32kv_rel=worst_spl_f32kv / scale, vit_err_split=e_vit_spl,
               vit_err_plain=e_vit_plain, vit_err_parity=e_vit_par,
               vit_out_absmax=float(y_ref.abs().max()), layernorm_outlier_channels=info["outliers"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"trained_like_logits_{head_scale:g}.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("\ntrained-like statistics: " + json.dumps(rec))
    assert worst_plain <= PLAIN_REL * max(1.0, scale), rec
    assert worst_par <= PARITY_REL * max(1.0, scale), rec
    assert worst_par <= 1e-3, rec                # the absolute tolerance of BASELINE.json, at every scale tested
    assert worst_spl <= SPLIT_REL * max(1.0, scale) and worst_spl_f32kv <= PARITY_REL * max(1.0, scale), rec
    assert worst_spl <= 1e-3, rec                # ... and in the parity-QUALIFIED configuration bench.py times (`parity_qualified`)


def test_free_running_captions_with_staggered_endings_equal_the_oracle_at_trained_like_statistics():
    """[SEP] biased so that hypotheses finish at different lengths (BeamHypotheses.is_done fires per image; finished
    images leave the decode batch: blip.py decode compaction) on the trained-like weights, parity mode, 6 images:
    token sequences equal the fp32 oracle's beam search."""
    from oracle import beam_ref, clip_ref, med_ref, vit_ref
    from vidil_amd.packing import set_compute_dtype, set_parity_mode

    cap, sd, _ = _build(2.0, sep_bias=12.0)       # (tuned on the CPU oracle: one search runs to the length limit, five end at 6 tokens)
    cap = cap.to(DEV)
    set_compute_dtype("f16", cap)
    set_parity_mode(True, cap)
    B, nb = 6, 3
    u8 = synthetic_frames(1, B, first_video=33)[0]
    with torch.no_grad():
        y_ref = vit_ref.vit_forward(sd, clip_ref.preprocess_u8(u8))
    enc3 = y_ref.repeat_interleave(nb, dim=0)
    state = {}

    def step(ids, beam_idx):
        with torch.no_grad():
            past = None if beam_idx is None else med_ref.reorder_cache(state["cache"], torch.from_numpy(beam_idx))
            lg, state["cache"] = med_ref.decoder_logits(sd, torch.from_numpy(ids), enc3, past)
        return lg.numpy()

    prompt = cap.prompt_ids(B, "cpu").long().numpy()
    otrace = []
    seqs, _ = beam_ref.beam_search(step, prompt, num_beams=nb, max_length=20, min_length=5, eos_token_id=102, pad_token_id=0, trace=otrace)
    _, y3 = cap.visual_encoder.forward_u8(torch.from_numpy(u8).to(DEV), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
    out_tok, out_len = cap.generate_ids(y3, B, num_beams=nb, max_length=20, min_length=5)
    toks = out_tok.cpu().numpy()
    lens = sorted(len(s) for s in seqs)
    gaps = np.stack([np.min(t["cand_scores"][:, :-1] - t["cand_scores"][:, 1:], axis=1) for t in otrace if t["cand_scores"].shape[0] == B]) \
        if otrace else np.zeros((0, B))
    agree = [bool(np.array_equal(toks[b][: len(seqs[b])], seqs[b])) for b in range(B)]
    print(f"\ntrained-like, staggered endings: oracle caption lengths {lens}; device == oracle for {sum(agree)}/{B} images")
    assert len(set(lens)) > 1, "the [SEP] bias should make searches end at different lengths"
    # every image whose oracle decisions all had a margin must agree; a near-tie (< 1e-3 between candidates) may flip
    for b in range(B):
        decisive = gaps.shape[0] == 0 or gaps[:, b].min() > 1e-3
        assert agree[b] or not decisive, (b, seqs[b], toks[b])
    assert sum(agree) >= B - 1


def test_layernorm_fold_vs_unfolded_tower_on_trained_like_weights():
    """The LN-folded tower (raw x rounded to 16 bits, normalisation applied to the accumulators) against the unfolded
    kernels (LayerNorm kernel + plain GEMM) on weights with outlier gains and off-zero row means, both against the fp32
    oracle: the fold may cost at most FOLD_BUDGET x the unfolded path's error (sqrt(1 + (mu/sigma)^2) per GEMM)."""
    from oracle import clip_ref, vit_ref
    from vidil_amd.packing import set_compute_dtype

    cap, sd, _ = _build(1.0)
    vit = cap.visual_encoder.to(DEV)
    set_compute_dtype("f16", vit)
    u8 = synthetic_frames(1, 3, first_video=41)[0]
    with torch.no_grad():
        y_ref = vit_ref.vit_forward({k: v for k, v in sd.items()}, clip_ref.preprocess_u8(u8))
    errs = {}
    for fuse in (True, False):
        vit.fuse_layernorm = fuse
        y32, _ = vit.forward_u8(torch.from_numpy(u8).to(DEV), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
        d = (y32.cpu() - y_ref).abs()
        errs[fuse] = (d.max().item(), d.mean().item())
    scale = float(y_ref.abs().max())
    print(f"\ntrained-like ViT-B/16 (|y|max {scale:.1f}): folded LN max {errs[True][0]:.2e} mean {errs[True][1]:.2e}; "
          f"unfolded max {errs[False][0]:.2e} mean {errs[False][1]:.2e}")
    FOLD_BUDGET = 2.0
    assert errs[True][1] <= FOLD_BUDGET * errs[False][1] + 1e-6
    assert errs[True][0] <= 2e-2 * max(1.0, scale / 10.0)
