"""BLIP image-text-matching filter on the HIP kernels — drop-in for the reference's
``models/blip_itm.py`` on the hot path (``blip_itm(...)``, ``BLIP_ITM.forward(image,
caption, match_head='itm') -> [F,2]`` raw logits).

``itm_pairs`` is the de-duplicated schedule CapFilt uses: the filter ViT and the
per-layer cross-attention K/V run ONCE per frame, and every (frame, caption) pair of
a batch of videos goes through the text encoder in one pass — the reference instead
re-runs the whole ViT once per caption (run_video_CapFilt.py:110-112 ->
models/blip_itm.py:43).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import kernels as K
from .blip import create_vit, load_checkpoint, resolve_med_config
from .med import BertConfig, BertModel
from .packing import PackedCache, require_cuda, v32, w3, w16
from .tokenizer import init_tokenizer, refuse_synthetic_with_checkpoint

ITM_MAX_LENGTH = 35  # models/blip_itm.py:46


class BLIP_ITM(PackedCache, nn.Module):
    def __init__(self, med_config="configs/med_config.json", image_size=384, vit="base", vit_grad_ckpt=False,
                 vit_ckpt_layer=0, embed_dim=256, tokenizer=None):
        super().__init__()
        self.visual_encoder, vision_width = create_vit(vit, image_size, vit_grad_ckpt, vit_ckpt_layer)
        self.tokenizer = tokenizer if tokenizer is not None else init_tokenizer()
        cfg = BertConfig.from_json_file(resolve_med_config(med_config))
        cfg.encoder_width = vision_width
        self.text_encoder = BertModel(config=cfg, add_pooling_layer=False)
        text_width = cfg.hidden_size
        self.vision_proj = nn.Linear(vision_width, embed_dim)   # 'itc' head (forward(match_head='itc'))
        self.text_proj = nn.Linear(text_width, embed_dim)
        self.itm_head = nn.Linear(text_width, 2)

    def _pack(self):
        p = dict(itm_w=w16(self.itm_head.weight, dtype=self.cdt), itm_b=v32(self.itm_head.bias), parity=self.parity)
        if p["parity"]:          # parity precision mode: [W_hi | W_hi | W_lo] against the [hi | lo | hi] rows of the [CLS] states
            p["itm_w3"] = w3(self.itm_head.weight, dtype=self.cdt)
        # the 'itc' head's two projections (models/blip_itm.py:60-67; plain operands)
        p.update(vp_w=w16(self.vision_proj.weight, dtype=self.cdt), vp_b=v32(self.vision_proj.bias),
                 tp_w=w16(self.text_proj.weight, dtype=self.cdt), tp_b=v32(self.text_proj.bias))
        return p

    def parameters_for_fingerprint(self):
        return [self.itm_head.weight, self.itm_head.bias, self.vision_proj.weight, self.vision_proj.bias,
                self.text_proj.weight, self.text_proj.bias]

    def _project_cls(self, h16, rows, T, w, b):
        """normalize(Linear(token 0 of each of ``rows`` sequences of length T)) -> f32 [rows, embed_dim]."""
        C = h16.shape[-1]
        out = torch.empty((rows, w.shape[0]), dtype=torch.float32, device=h16.device)
        K.gemm(h16.view(-1), w, b, out=out, M=rows, lda=T * C)
        return K.l2_normalize_rows(out)

    @torch.no_grad()
    def itc_similarity(self, y16, n_images, captions, device):
        """models/blip_itm.py:60-67 (match_head='itc'): normalize(vision_proj(image [CLS])) @ normalize(text_proj(text [CLS]))^T
        with the text encoder in mode='text' (no cross-attention) -> f32 [n_images, len(captions)], exact-f32 scores."""
        if self.parity:
            raise NotImplementedError("match_head='itc' has no parity-precision form (it is not on the CapFilt path)")
        p = self.packed()
        img = self._project_cls(y16, n_images, y16.shape[0] // n_images, p["vp_w"], p["vp_b"])
        ids, lens = self.tokenize(list(captions))
        t_eff = max(1, min(ITM_MAX_LENGTH, int(lens.max().item())))
        d_ids = ids[:, :t_eff].to(device).contiguous()
        _, h16 = self.text_encoder.encode(d_ids, lens.to(device).contiguous(), None)
        txt = self._project_cls(h16, d_ids.shape[0], t_eff, p["tp_w"], p["tp_b"])
        return K.scan_scores(img, txt)

    # ------------------------------------------------------------------ tokenisation
    def tokenize(self, captions):
        """models/blip_itm.py:46-47: padding='max_length', truncation, max_length=35; first id stays [CLS]."""
        enc = self.tokenizer(captions, padding="max_length", truncation=True, max_length=ITM_MAX_LENGTH,
                             return_tensors="pt")
        ids = enc.input_ids.to(torch.int32)
        lens = enc.attention_mask.sum(dim=1).to(torch.int32)
        return ids, lens

    # ------------------------------------------------------------------ de-duplicated schedule
    @torch.no_grad()
    def project_image_kv(self, enc16, n_images, min_rows_per_image):
        """The cross-attention K/V of every layer for a batch of images, for one or more ``itm_pairs`` calls on the same
        frames (``cross=``).  ``min_rows_per_image``: the smallest number of query rows a launch will present per
        image — row-major V (cheaper stores) is only readable by the staged kernel, i.e. above 32 rows."""
        Te = enc16.shape[0] // n_images
        return self.text_encoder.project_cross_kv(enc16, n_images, Te, v_rowmajor=min_rows_per_image > 32,
                                                  last_layer_vt=True)

    @torch.no_grad()
    def itm_pairs(self, enc16, n_images, ids, lens, image_index=None, group_start=None, max_group=0, pair_text=None,
                  cross=None):
        """enc16 f16 [n_images*Te, width]; ids i32 [P,35]; lens i32 [P].  Either image_index i32 [P] (pair ->
        image, any order) or, for IMAGE-MAJOR pair order, group_start i32 [n_images+1] (pairs of image j are
        group_start[j] .. group_start[j+1]-1, at most max_group of them), which lets one fetch of an image's
        cross K/V serve all its captions.  pair_text (int [P]): ids / lens then hold the U DISTINCT texts and pair p
        scores text pair_text[p] — the text-only front of the encoder runs once per text (BertModel.encode_cls).
        cross: ``project_image_kv``'s result for these images when several calls score pairs of the same frames.
        Returns f32 [P,2] raw ITM logits."""
        require_cuda(enc16, "BLIP_ITM")
        te = self.text_encoder
        dev = enc16.device
        Te = enc16.shape[0] // n_images
        # The reference pads every caption to 35 tokens (models/blip_itm.py:46).  Padded keys are masked and a
        # padded row never feeds a real one, so the [CLS] output is unchanged if the batch is cut to its
        # longest real caption; this removes the all-padding columns.
        t_eff = min(ids.shape[1], int(lens.max().item())) if ids.shape[0] else ids.shape[1]
        # image-major groups with more than 32 query rows go through the staged attention kernel, which takes V
        # row-major (plain 16-B stores from the K|V GEMM instead of the V^T scatter)
        rows_per_image = (max_group if group_start is not None else 1) * t_eff
        if cross is None:
            cross = te.project_cross_kv(enc16, n_images, Te, v_rowmajor=rows_per_image > 32, last_layer_vt=True)
        elif cross.NP == 0 and rows_per_image <= 32:
            raise K.VidilHipError("itm_pairs: cross= holds row-major values but this call has at most 32 query rows "
                                  "per image (project_image_kv(min_rows_per_image=...))")
        ids = ids[:, :t_eff].to(dev).contiguous()
        lens = lens.to(dev).contiguous()
        if group_start is not None:
            group_start = group_start.to(dev).to(torch.int32).contiguous()
        else:
            image_index = image_index.to(dev).to(torch.int32).contiguous()
        if pair_text is not None:
            pair_text = pair_text.to(dev).to(torch.int64).contiguous()
        P = ids.shape[0] if pair_text is None else pair_text.numel()
        p = self.packed()
        if p["parity"]:
            # parity precision mode (packing.set_parity_mode): every pair through ALL layers on all of its tokens with
            # error-compensated GEMM operands (BertModel.encode -> _run_layers_parity) — no [CLS]-only last layer, no
            # shared text front: a statement about results, not the throughput schedule
            if pair_text is not None:
                ids, lens = ids.index_select(0, pair_text).contiguous(), lens.index_select(0, pair_text).contiguous()
            h32, _ = te.encode(ids, lens, cross, cross_index=image_index, cross_groups=group_start, cross_max_group=max_group)
            C = te.config.hidden_size
            cls32 = h32.view(P, ids.shape[1], C)[:, 0].contiguous()
            a3 = K.split3(cls32, torch.empty((P, 3 * C), dtype=p["itm_w"].dtype, device=dev))
            out = torch.empty((P, 2), dtype=torch.float32, device=dev)
            K.gemm(a3, p["itm_w3"], p["itm_b"], out=out, split_k=True)
            return out
        # only token 0 feeds the itm_head: the last layer runs on the [CLS] rows alone (BertModel.encode_cls)
        _, c16 = te.encode_cls(ids, lens, cross, cross_index=image_index, cross_groups=group_start,
                               cross_max_group=max_group, pair_text=pair_text)
        out = torch.empty((P, 2), dtype=torch.float32, device=dev)
        K.gemm(c16, p["itm_w"], p["itm_b"], out=out)
        return out

    @torch.no_grad()
    def forward(self, image, caption, match_head="itm"):
        """Reference call shape (models/blip_itm.py:41-58): F images, F captions -> [F,2]."""
        if match_head not in ("itm", "itc"):
            raise ValueError(f"unknown match_head {match_head!r}")
        require_cuda(image, "BLIP_ITM.forward")
        F = image.shape[0]
        _, y16 = self.visual_encoder.forward_both(image)
        if match_head == "itc":          # [F, F] similarities (models/blip_itm.py:60-67)
            return self.itc_similarity(y16, F, caption, image.device)
        ids, lens = self.tokenize(list(caption))
        return self.itm_pairs(y16, F, ids, lens, torch.arange(F, dtype=torch.int32))


def blip_itm(pretrained="", **kwargs):
    """Reference: models/blip_itm.py:70-75."""
    model = BLIP_ITM(**kwargs)
    if pretrained:
        refuse_synthetic_with_checkpoint(model.tokenizer, pretrained)
        model, msg = load_checkpoint(model, pretrained)
        assert len(msg.missing_keys) == 0
    return model
