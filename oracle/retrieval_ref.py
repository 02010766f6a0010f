"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement (fp32 PyTorch) of the BLIP retrieval backend of
the visual tokenizer, ``--encoder_version blip``:

  * get_image_embeddings_blip   run_visual_tokenization.py:152-159  (ViT -> vision_proj(cls) -> normalize)
  * get_text_embeddings_blip    run_visual_tokenization.py:113-133  (text_encoder mode='text' -> text_proj(cls) ->
                                                                     normalize; ids[:,0] := [ENC] afterwards)
  * the per-frame ITM re-rank   run_visual_tokenization.py:277-293  (topk(k_test) of the similarities, text_encoder
                                                                     with the frame's image tokens, itm_head[:,1] + sim,
                                                                     -100 elsewhere)
on top of oracle/vit_ref.py and oracle/med_ref.py (which are pinned against the reference's models/vit.py and
models/med.py).  The glue above is a few lines of the reference's driver and is restated, not imported (the driver
needs decord / spacy / ruamel and an HTTP download at import time)."""
import torch
import torch.nn.functional as F

from . import med_ref, vit_ref


def image_features(sd, x, *, depth=12, heads=12, patch=16):
    y = vit_ref.vit_forward(sd, x, depth=depth, heads=heads, patch=patch)
    emb = F.normalize(F.linear(y[:, 0, :], sd["vision_proj.weight"], sd["vision_proj.bias"]), dim=-1)
    return y, emb


def text_features(sd, ids, mask, *, layers=12, H=12):
    h, _ = med_ref.bert_model(sd, "text_encoder.", ids, mask, layers=layers, H=H, enc=None, is_decoder=False)
    return F.normalize(F.linear(h[:, 0, :], sd["text_proj.weight"], sd["text_proj.bias"]), dim=-1)


def score_matrix(sd, image_feats, image_embeds, text_embeds, ids_enc, mask, k_test, *, layers=12, H=12):
    """[F, N] matrix of run_visual_tokenization.py:277-293 for one category."""
    sims = image_embeds @ text_embeds.t()
    score = torch.full_like(sims, -100.0)
    for i in range(sims.shape[0]):
        topk_sim, topk_idx = sims[i].topk(k=k_test, dim=0)
        enc = image_feats[i].repeat(k_test, 1, 1)
        h, _ = med_ref.bert_model(sd, "text_encoder.", ids_enc[topk_idx], mask[topk_idx], layers=layers, H=H, enc=enc,
                                  is_decoder=False)
        s = F.linear(h[:, 0, :], sd["itm_head.weight"], sd["itm_head.bias"])[:, 1]
        score[i, topk_idx] = s + topk_sim
    return sims, score
