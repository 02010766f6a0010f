"""BLIP caption decoder on the HIP kernels — drop-in for the reference's
``models/blip.py`` on the hot path: ``blip_decoder(pretrained, image_size, vit)``,
``BLIP_Decoder.generate(image, sample=False, num_beams=3, max_length=…, min_length=…)
-> list[str]``, ``init_tokenizer``, ``create_vit``, ``load_checkpoint``.

The whole beam search (log-softmax, EOS ban, top-2k, BeamSearchScorer bookkeeping,
KV-cache reorder) runs on the device; the host only enqueues kernels and reads the
final token ids back once per batch.
"""
from __future__ import annotations

import os
from urllib.parse import urlparse

import torch
import torch.nn as nn

from . import kernels as K
from .med import BertConfig, BertLMHeadModel
from .packing import require_cuda
from .tokenizer import init_tokenizer  # noqa: F401  (re-exported, reference API)
from .vit import VisionTransformer, interpolate_pos_embed

_DEFAULT_MED_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "med_config.json")

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # run_video_CapFilt.py:133
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def resolve_med_config(path):
    """The reference default is the repo-relative 'configs/med_config.json'; fall back to the packaged copy."""
    if path and os.path.isfile(path):
        return path
    return _DEFAULT_MED_CONFIG


def create_vit(vit, image_size, use_grad_checkpointing=False, ckpt_layer=0, drop_path_rate=0, **_):
    """Reference: models/blip.py:298-326 (widths/depths/heads of 'base' and 'large')."""
    if vit == "base":
        width = 768
        enc = VisionTransformer(img_size=image_size, patch_size=16, embed_dim=width, depth=12, num_heads=12)
    elif vit == "large":
        width = 1024
        enc = VisionTransformer(img_size=image_size, patch_size=16, embed_dim=width, depth=24, num_heads=16)
    else:
        raise ValueError("cannot create vit:", vit)
    return enc, width


def is_url(url_or_filename):
    return urlparse(url_or_filename).scheme in ("http", "https")


def load_checkpoint(model, url_or_filename):
    """Reference semantics (models/blip.py:332-354): checkpoint['model'], position-embedding
    interpolation, shape-mismatched keys dropped, strict=False."""
    if is_url(url_or_filename):
        from torch.hub import load_state_dict_from_url

        checkpoint = load_state_dict_from_url(url_or_filename, map_location="cpu", check_hash=False, progress=True)
    elif os.path.isfile(url_or_filename):
        checkpoint = torch.load(url_or_filename, map_location="cpu")
    else:
        raise RuntimeError("checkpoint url or path is invalid")
    state_dict = checkpoint["model"]
    state_dict["visual_encoder.pos_embed"] = interpolate_pos_embed(state_dict["visual_encoder.pos_embed"],
                                                                   model.visual_encoder)
    own = model.state_dict()
    for key in list(own.keys()):
        if key in state_dict and state_dict[key].shape != own[key].shape:
            del state_dict[key]
    msg = model.load_state_dict(state_dict, strict=False)
    print("load checkpoint from %s" % url_or_filename)
    return model, msg


class DecodeTrace:
    """Optional capture of per-step tensors for parity tests (off on the hot path)."""

    def __init__(self):
        self.logits = []       # f32 [R,V] per forward
        self.cand_scores = []
        self.cand_index = []


class BLIP_Decoder(nn.Module):
    def __init__(self, med_config="configs/med_config.json", image_size=384, vit="base", vit_grad_ckpt=False,
                 vit_ckpt_layer=0, prompt="a picture of ", tokenizer=None):
        super().__init__()
        self.visual_encoder, vision_width = create_vit(vit, image_size, vit_grad_ckpt, vit_ckpt_layer)
        self.tokenizer = tokenizer if tokenizer is not None else init_tokenizer()
        cfg = BertConfig.from_json_file(resolve_med_config(med_config))
        cfg.encoder_width = vision_width
        self.text_decoder = BertLMHeadModel(config=cfg)
        self.prompt = prompt
        self.prompt_length = len(self.tokenizer(self.prompt).input_ids) - 1

    # ------------------------------------------------------------------ prompt ids
    def prompt_ids(self, B, device):
        """models/blip.py:135-138: tokenise the prompt, first id := [DEC], drop the trailing [SEP]."""
        ids = self.tokenizer([self.prompt] * B, return_tensors="pt").input_ids
        ids[:, 0] = self.tokenizer.bos_token_id
        return ids[:, :-1].to(torch.int32).to(device)

    # ------------------------------------------------------------------ beam decode
    @torch.no_grad()
    def generate_ids(self, enc16, B, *, num_beams=3, max_length=30, min_length=10, trace: DecodeTrace = None,
                     check_done_every=4):
        """enc16: f16 [B*Te, width] image tokens of B images.  Returns (tokens i32 [B,max_length], lens i32 [B]):
        best hypothesis incl. the prompt, then [SEP] if it fits, then [PAD]."""
        require_cuda(enc16, "BLIP_Decoder.generate")
        dec, bert = self.text_decoder, self.text_decoder.bert
        cfg = dec.config
        tok = self.tokenizer
        eos, pad = tok.sep_token_id, tok.pad_token_id
        dev = enc16.device
        nb, R = num_beams, B * num_beams
        H, L, V = cfg.num_attention_heads, cfg.num_hidden_layers, cfg.vocab_size
        Te = enc16.shape[0] // B
        cross = bert.project_cross_kv(enc16, B, Te)
        Tcap = max_length
        NPs = (Tcap + 7) // 8 * 8
        kc = [torch.empty((L, R, H, Tcap, 64), dtype=torch.float16, device=dev) for _ in range(2)]
        vc = [torch.empty((L, R, H, 64, NPs), dtype=torch.float16, device=dev) for _ in range(2)]
        bufs = K.BeamBuffers(B, nb, max_length, dev)
        prompt = self.prompt_ids(B, dev)
        P = prompt.shape[1]
        bufs.reset(prompt)
        cur = 0
        ws = {}
        # ---- prefill over the prompt (all beams of an image start identical)
        ids = bufs.seqs[:, :P].contiguous().view(-1)
        h32, h16 = bert.embed(ids, P, 0)
        bert.run_layers(h32, h16, rows=R, T=P, self_k=kc[cur], self_vt=vc[cur], t_off=0, Tk_cap=Tcap, NPs=NPs,
                        causal=True, kv_len=None, cross=cross, cross_group=nb, ws=ws)
        logits = dec.lm_logits(h16, R, P)
        cur_len = P
        ws1 = {}
        while True:
            ban = eos if cur_len < min_length else -1
            cs, ci = K.logsoftmax_topk(logits, bufs.beam_scores, B, nb, ban)
            if trace is not None:
                trace.logits.append(logits.clone())
                trace.cand_scores.append(cs.clone())
                trace.cand_index.append(ci.clone())
            K.beam_update(bufs, cs, ci, V, cur_len, eos, pad)
            cur_len += 1
            if cur_len >= max_length:
                break
            if check_done_every and (cur_len % check_done_every == 0) and int(bufs.n_done.item()) == B:
                break
            K.kv_reorder(kc[cur], kc[cur ^ 1], bufs.beam_idx, L, R)
            K.kv_reorder(vc[cur], vc[cur ^ 1], bufs.beam_idx, L, R)
            cur ^= 1
            h32, h16 = bert.embed(bufs.next_tok, 1, cur_len - 1)
            bert.run_layers(h32, h16, rows=R, T=1, self_k=kc[cur], self_vt=vc[cur], t_off=cur_len - 1, Tk_cap=Tcap,
                            NPs=NPs, causal=False, kv_len=None, cross=cross, cross_group=nb, ws=ws1)
            logits = dec.lm_logits(h16, R, 1, out=logits)
        out_tok, out_len, _ = K.beam_finalize(bufs, cur_len, eos, pad)
        return out_tok, out_len

    def decode_captions(self, out_tok):
        captions = []
        for row in out_tok.cpu().tolist():
            text = self.tokenizer.decode(row, skip_special_tokens=True)
            captions.append(text[len(self.prompt):])
        return captions

    @torch.no_grad()
    def generate(self, image, sample=False, num_beams=3, max_length=30, min_length=10, top_p=0.9,
                 repetition_penalty=1.0):
        """Reference: models/blip.py:127-167.  image f32 [B,3,S,S] on the GPU -> list of B captions."""
        if sample:
            raise NotImplementedError("nucleus sampling (generation_mode != 'beam') is not built yet; see DESIGN.md §next")
        if repetition_penalty != 1.0:
            raise NotImplementedError("repetition_penalty != 1.0 is not on the hot path")
        _, y16 = self.visual_encoder.forward_both(image)
        out_tok, _ = self.generate_ids(y16, image.shape[0], num_beams=num_beams, max_length=max_length,
                                       min_length=min_length)
        return self.decode_captions(out_tok)

    def forward(self, image, caption):
        raise NotImplementedError("training loss is out of scope (inference hot path only)")


def blip_decoder(pretrained="", **kwargs):
    """Reference: models/blip.py:269-274."""
    model = BLIP_Decoder(**kwargs)
    if pretrained:
        model, msg = load_checkpoint(model, pretrained)
        assert len(msg.missing_keys) == 0
    return model
