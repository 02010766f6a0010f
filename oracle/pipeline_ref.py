"""oracle/pipeline_ref.py — TEST INFRASTRUCTURE.  The reference's per-video hot loops
restated end to end on the CPU in fp32, IN THE REFERENCE'S SCHEDULE (ViT re-run per
caption by the filter, cross-attention K/V re-projected on every decoder call, full
score matrix + argsort for the visual tokens).  Used as the end-to-end parity checker
and as ``bench.py``'s ``cpu_baseline`` (kind "port").

Follows run_video_CapFilt.py:155-204 and run_visual_tokenization.py:226-312.
"""
from __future__ import annotations

import numpy as np
import torch

from . import beam_ref, clip_ref, med_ref, tokens_ref, vit_ref


def caption_video(sd_cap, frames_f32, prompt_ids, tokenizer, prompt_text, *, depth=12, heads=12,
                  num_beams=3, max_length=20, min_length=5, trace=None, dedup=False):
    """BLIP_Decoder.generate (models/blip.py:127-167) for the F frames of one video -> list of F strings.
    dedup=True (bench.py's second CPU baseline, same results): the cross-attention K/V of every layer are projected
    once per search instead of on every decoder call (the image rows of a beam never change)."""
    F = frames_f32.shape[0]
    with torch.no_grad():
        enc = vit_ref.vit_forward(sd_cap, frames_f32, prefix="visual_encoder.", depth=depth, heads=heads)
        enc = enc.repeat_interleave(num_beams, dim=0)          # models/blip.py:130
    state = {}
    cross_cache = {} if dedup else None

    def step(ids, beam_idx):
        ids_t = torch.from_numpy(ids)
        with torch.no_grad():
            past = None if beam_idx is None else med_ref.reorder_cache(state["cache"], torch.from_numpy(beam_idx))
            logits, cache = med_ref.decoder_logits(sd_cap, ids_t, enc, past, cross_cache=cross_cache)
        state["cache"] = cache
        return logits.numpy()

    prompts = np.repeat(np.asarray(prompt_ids, dtype=np.int64)[None], F, axis=0)
    seqs, _ = beam_ref.beam_search(step, prompts, num_beams=num_beams, max_length=max_length, min_length=min_length,
                                   eos_token_id=tokenizer.sep_token_id, pad_token_id=tokenizer.pad_token_id, trace=trace)
    caps = []
    for s in seqs:
        text = tokenizer.decode(s.tolist(), skip_special_tokens=True)
        caps.append(text[len(prompt_text):])
    return caps


def filter_video(sd_itm, frames_f32, captions, tokenizer, threshold, mode="max_filter", *, depth=12, heads=12,
                 return_probs=False, dedup=False):
    """filter_captions (run_video_CapFilt.py:107-126): one BLIP_ITM.forward per caption, each re-running the ViT.
    dedup=True: the ViT and the cross-attention K/V run once per frame and serve every caption (same results)."""
    F = frames_f32.shape[0]
    kept, probs = [], []
    enc, cross_cache = None, ({} if dedup else None)
    for t in captions:
        with torch.no_grad():
            if enc is None or not dedup:
                enc = vit_ref.vit_forward(sd_itm, frames_f32, prefix="visual_encoder.", depth=depth, heads=heads)
            tk = tokenizer([t] * F, padding="max_length", truncation=True, max_length=35, return_tensors="pt")
            out = med_ref.itm_logits(sd_itm, enc, tk.input_ids, tk.attention_mask, cross_cache=cross_cache)
            score = med_ref.filter_scores(out).numpy()
        probs.append(score)
        if tokens_ref.keep_caption(score, threshold, mode):
            kept.append(t)
    return (kept, probs) if return_probs else kept


def capfilt_video(sd_cap, sd_itm, frames_f32, prompt_ids, tokenizer, prompt_text, threshold=0.4, dedup=False, **kw):
    """One iteration of the CapFilt hot loop with caption=True, filter=True, filter_generated_only=True,
    keep_original_caption=False (the shipped configs).  Returns (filtered, unfiltered)."""
    caps = tokens_ref.dedup_captions(caption_video(sd_cap, frames_f32, prompt_ids, tokenizer, prompt_text, dedup=dedup, **kw))
    kept = filter_video(sd_itm, frames_f32, caps, tokenizer, threshold, dedup=dedup)
    return kept, caps


def visual_tokens_video(sd_clip, frames_f32, text_embeds_by_cat, texts_by_cat, topk=5, *, layers=12, heads=12, patch=32):
    """One video of predict_video (run_visual_tokenization.py:226-312): image embeds, full score
    matrices, argsort top-k, aggregation."""
    with torch.no_grad():
        emb = clip_ref.image_embeds(sd_clip, frames_f32, layers=layers, heads=heads, patch=patch)
        scores = {k: (emb @ text_embeds_by_cat[k].t()).numpy() for k in tokens_ref.CATEGORIES}
    return tokens_ref.visual_tokens_from_scores(["v"], [[]], scores, texts_by_cat, frames_f32.shape[0], topk)["v"]
