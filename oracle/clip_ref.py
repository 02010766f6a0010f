"""oracle/clip_ref.py — TEST INFRASTRUCTURE.  fp32 restatement of the CLIP towers
the reference uses through HuggingFace ``transformers.CLIPModel``
(run_visual_tokenization.py:9,83-96,135-143,347-350).  CLIP is third-party code
not vendored in the reference tree; this restates the published architecture
(pre-LN transformer, quick-GELU, class token + learned positions, ``pre_layrnorm``
/ ``post_layernorm`` on the vision side, causal text tower pooled at the first
EOS token, bias-free projections, L2-normalised outputs) and is pinned against
the installed ``transformers`` (5.15) ``CLIPModel`` in tests/test_oracle_vs_reference.py.

State-dict keys are HF's (``vision_model.encoder.layers.0.self_attn.q_proj.weight`` …).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LN_EPS = 1e-5
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], LN_EPS)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def encoder_layer(sd, p, x, heads, add_mask=None):
    B, T, C = x.shape
    hd = C // heads
    h = _ln(sd, p + "layer_norm1", x)
    q = _lin(sd, p + "self_attn.q_proj", h).view(B, T, heads, hd).transpose(1, 2)
    k = _lin(sd, p + "self_attn.k_proj", h).view(B, T, heads, hd).transpose(1, 2)
    v = _lin(sd, p + "self_attn.v_proj", h).view(B, T, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    if add_mask is not None:
        s = s + add_mask
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, T, C)
    x = x + _lin(sd, p + "self_attn.out_proj", a)
    h = _ln(sd, p + "layer_norm2", x)
    h = _lin(sd, p + "mlp.fc2", quick_gelu(_lin(sd, p + "mlp.fc1", h)))
    return x + h


def preprocess_u8(frames_u8):
    """HF CLIPImageProcessor for frames already S x S (resize/crop are identity):
    uint8 [F,S,S,3] -> f32 [F,3,S,S], x/255 then (x-mean)/std.  The reference's BLIP
    preprocessing (run_video_CapFilt.py:128-137) uses the same constants."""
    x = torch.as_tensor(frames_u8).permute(0, 3, 1, 2).to(torch.float32) / 255.0
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (x - mean) / std


def pooled_output(sd, pixel_values, *, layers=12, heads=12, patch=32):
    """CLIPModel(...).vision_model_output.pooler_output: post_layernorm of the class token, [F,3,S,S] -> [F,D]
    (what data/video_pretrain_dataset.py:199-202 clusters for 'clip-kmeans')."""
    p = "vision_model."
    B = pixel_values.shape[0]
    x = F.conv2d(pixel_values, sd[p + "embeddings.patch_embedding.weight"], None, stride=patch)
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[p + "embeddings.position_embedding.weight"][None, : x.shape[1] + 1]
    x = _ln(sd, p + "pre_layrnorm", x)
    for i in range(layers):
        x = encoder_layer(sd, f"{p}encoder.layers.{i}.", x, heads)
    return _ln(sd, p + "post_layernorm", x[:, 0])


def image_embeds(sd, pixel_values, *, layers=12, heads=12, patch=32):
    """CLIPModel(...).image_embeds: [F,3,S,S] -> unit-norm [F,P]."""
    p = "vision_model."
    B = pixel_values.shape[0]
    x = F.conv2d(pixel_values, sd[p + "embeddings.patch_embedding.weight"], None, stride=patch)
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[p + "embeddings.position_embedding.weight"][None, : x.shape[1] + 1]
    x = _ln(sd, p + "pre_layrnorm", x)
    for i in range(layers):
        x = encoder_layer(sd, f"{p}encoder.layers.{i}.", x, heads)
    pooled = _ln(sd, p + "post_layernorm", x[:, 0])
    e = F.linear(pooled, sd["visual_projection.weight"])
    return e / e.norm(p=2, dim=-1, keepdim=True)


def text_embeds(sd, input_ids, attention_mask=None, *, layers=12, heads=8, eos_token_id=49407):
    """CLIPModel(...).text_embeds: ids [T,L] -> unit-norm [T,P]; pooled at the first EOS."""
    p = "text_model."
    B, L = input_ids.shape
    x = sd[p + "embeddings.token_embedding.weight"][input_ids] + sd[p + "embeddings.position_embedding.weight"][None, :L]
    neg = torch.finfo(torch.float32).min
    mask = torch.full((L, L), neg).triu(1)[None, None]
    if attention_mask is not None:
        pad = (1.0 - attention_mask.to(torch.float32))[:, None, None, :] * neg
        mask = torch.clamp(mask + pad, min=neg)
    for i in range(layers):
        x = encoder_layer(sd, f"{p}encoder.layers.{i}.", x, heads, mask)
    x = _ln(sd, p + "final_layer_norm", x)
    if eos_token_id == 2:
        pos = input_ids.argmax(dim=-1)
    else:
        pos = (input_ids == eos_token_id).int().argmax(dim=-1)
    pooled = x[torch.arange(B), pos]
    e = F.linear(pooled, sd["text_projection.weight"])
    return e / e.norm(p=2, dim=-1, keepdim=True)
