"""BLIP retrieval model on the HIP kernels — the ``--encoder_version blip`` backend of the visual tokenizer
(run_visual_tokenization.py:113-133,152-159,277-293; reference model: models/blip_retrieval.py:14-75).

Inference only: the momentum encoders, the feature queues and the contrastive / matching losses of the reference
class are training state; their checkpoint keys are ignored by ``load_checkpoint`` (``strict=False``, as in the
reference's own factory, which only prints the missing keys).  What the tokenizer uses:

  * ``image_features_u8(frames)``  -> ViT tokens (f16, for the ITM cross-attention) and the unit-norm ITC embedding
    ``normalize(vision_proj(cls))``                                   (get_image_embeddings_blip, :152-159)
  * ``text_features(texts)``       -> unit-norm ``normalize(text_proj(text_encoder(ids, mode='text')[:,0]))`` plus the
    ids (first token replaced by [ENC]) and lengths for the re-rank    (get_text_embeddings_blip, :113-133)
  * ``rerank(...)``                -> ``itm_head(text_encoder(ids, image)[:,0])[:,1]`` for (frame, text) pairs (:283-292)
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import kernels as K
from .blip import CLIP_MEAN, CLIP_STD, load_checkpoint
from .blip_itm import ITM_MAX_LENGTH, BLIP_ITM
from .packing import require_cuda, v32, w16
from .tokenizer import refuse_synthetic_with_checkpoint


class BLIP_Retrieval(BLIP_ITM):
    def __init__(self, med_config="configs/med_config.json", image_size=384, vit="base", vit_grad_ckpt=False,
                 vit_ckpt_layer=0, embed_dim=256, queue_size=57600, momentum=0.995, negative_all_rank=False,
                 tokenizer=None):
        super().__init__(med_config=med_config, image_size=image_size, vit=vit, vit_grad_ckpt=vit_grad_ckpt,
                         vit_ckpt_layer=vit_ckpt_layer, embed_dim=embed_dim, tokenizer=tokenizer)
        self.temp = nn.Parameter(0.07 * torch.ones([]))      # models/blip_retrieval.py:66 (unused at inference)
        self.queue_size, self.momentum, self.negative_all_rank = queue_size, momentum, negative_all_rank
        # the retrieval heads have no parity form: pin this model (and its towers) to the plain mode so that the process-wide
        # $VIDIL_PARITY default — meant for the captioner, the filter and CLIP — leaves it working
        from .packing import set_parity_mode
        set_parity_mode(False, self)

    def _pack(self):
        if self.__dict__.get("_parity"):     # (explicitly switched on for THIS model)
            raise NotImplementedError("the parity precision mode is built for BLIP_Decoder, BLIP_ITM and CLIPModel, not for the retrieval heads")
        return super()._pack()          # (BLIP_ITM packs the two ITC projections: vp_w / vp_b / tp_w / tp_b)

    # ------------------------------------------------------------------ features (_project_cls: BLIP_ITM)

    @torch.no_grad()
    def image_features_u8(self, frames_u8):
        """uint8 [N,S,S,3] -> (image tokens f16 [N*T, width], unit-norm embeddings f32 [N, embed_dim])."""
        require_cuda(frames_u8, "BLIP_Retrieval.image_features_u8")
        N = frames_u8.shape[0]
        _, y16 = self.visual_encoder.forward_u8(frames_u8, CLIP_MEAN, CLIP_STD)
        p = self.packed()
        return y16, self._project_cls(y16, N, y16.shape[0] // N, p["vp_w"], p["vp_b"])

    @torch.no_grad()
    def image_features(self, image):
        """f32 [N,3,S,S] (normalised) -> same as image_features_u8 (run_visual_tokenization.py:152-159)."""
        require_cuda(image, "BLIP_Retrieval.image_features")
        N = image.shape[0]
        _, y16 = self.visual_encoder.forward_both(image)
        p = self.packed()
        return y16, self._project_cls(y16, N, y16.shape[0] // N, p["vp_w"], p["vp_b"])

    @torch.no_grad()
    def text_features(self, texts, device, batch=512):
        """list[str] -> (unit-norm f32 [N,embed_dim], ids i32 [N,35] with the first id := [ENC], lens i32 [N])."""
        p = self.packed()
        embeds, all_ids, all_lens = [], [], []
        for i in range(0, len(texts), batch):
            ids, lens = self.tokenize(texts[i:i + batch])              # padding='max_length', max_length=35
            t_eff = max(1, min(ITM_MAX_LENGTH, int(lens.max().item())))
            d_ids = ids[:, :t_eff].to(device).contiguous()
            _, h16 = self.text_encoder.encode(d_ids, lens.to(device).contiguous(), None)     # mode='text': no cross-attn
            embeds.append(self._project_cls(h16, d_ids.shape[0], t_eff, p["tp_w"], p["tp_b"]))
            all_ids.append(ids)
            all_lens.append(lens)
        ids = torch.cat(all_ids, 0)
        ids[:, 0] = self.tokenizer.enc_token_id                         # run_visual_tokenization.py:132
        return torch.cat(embeds, 0), ids.to(device), torch.cat(all_lens, 0).to(device)

    @torch.no_grad()
    def rerank(self, y16, n_images, ids, lens, group_start, max_group):
        """ITM logit of class 1 for image-major (frame, text) pairs (see BLIP_ITM.itm_pairs) -> f32 [P]."""
        return self.itm_pairs(y16, n_images, ids, lens, group_start=group_start, max_group=max_group)[:, 1].contiguous()

    @torch.no_grad()
    def forward(self, image, caption, match_head="itm"):
        """models/blip_itm.py:41-67 call shape; 'itc' returns image_feat @ text_feat.t() [F,F]."""
        if match_head == "itm":
            return super().forward(image, caption, match_head)
        if match_head != "itc":
            raise ValueError(f"unknown match_head {match_head!r}")
        _, img = self.image_features(image)
        txt, _, _ = self.text_features(list(caption), image.device)
        return K.scan_scores(img, txt)


def blip_retrieval(pretrained="", **kwargs):
    """Reference: models/blip_retrieval.py:567-573 (prints, does not assert, the missing keys)."""
    model = BLIP_Retrieval(**kwargs)
    if pretrained:
        refuse_synthetic_with_checkpoint(model.tokenizer, pretrained)
        model, msg = load_checkpoint(model, pretrained)
        print("missing keys:")
        print(msg.missing_keys)
    return model
