"""oracle/med_ref.py — TEST INFRASTRUCTURE.  fp32 restatement of the reference's
MED (BERT with cross-attention), models/med.py, as used by the captioner
(BertLMHeadModel, models/blip.py:98,154-161) and the ITM filter (BertModel,
models/blip_itm.py:31,51-57).

State-dict keys are the reference's (``<prefix>encoder.layer.3.crossattention.self.key.weight`` …).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12  # configs/med_config.json:10


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], LN_EPS)


def embeddings(sd, p, ids, past_len=0):
    """models/med.py:71-94: word + position(past_len offset) -> LayerNorm."""
    T = ids.shape[1]
    x = sd[p + "word_embeddings.weight"][ids]
    x = x + sd[p + "position_embeddings.weight"][past_len: past_len + T][None]
    return _ln(sd, p + "LayerNorm", x)


def _heads(x, H):
    B, T, C = x.shape
    return x.view(B, T, H, C // H).permute(0, 2, 1, 3)


def self_attention(sd, p, h, add_mask, H, past_kv=None, kv_src=None, cross_cache=None):
    """models/med.py:143-225.  kv_src = encoder states for the cross branch (:160-163);
    past_kv = (k, v) cache concatenated on the sequence dim (:164-168).
    cross_cache (dict, optional; NOT the reference's schedule): keep the cross-branch K/V of this layer across calls
    — the reference recomputes them on every call; the de-duplicated CPU baseline of bench.py projects them once."""
    q = _heads(_lin(sd, p + "query", h), H)
    src = h if kv_src is None else kv_src
    if kv_src is not None and cross_cache is not None and p in cross_cache:
        k, v = cross_cache[p]
    else:
        k = _heads(_lin(sd, p + "key", src), H)
        v = _heads(_lin(sd, p + "value", src), H)
        if kv_src is not None and cross_cache is not None:
            cross_cache[p] = (k, v)
    if kv_src is None and past_kv is not None:
        k = torch.cat([past_kv[0], k], dim=2)
        v = torch.cat([past_kv[1], v], dim=2)
    s = q @ k.transpose(-1, -2)
    s = s / math.sqrt(q.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    pr = torch.softmax(s, dim=-1)
    ctx = (pr @ v).permute(0, 2, 1, 3).contiguous()
    return ctx.view(ctx.shape[0], ctx.shape[1], -1), (k, v)


def attention_block(sd, p, h, add_mask, H, past_kv=None, kv_src=None, cross_cache=None):
    """BertAttention = self + BertSelfOutput (dense, residual, LayerNorm) models/med.py:228-239,267-288."""
    ctx, kv = self_attention(sd, p + "self.", h, add_mask, H, past_kv, kv_src, cross_cache)
    out = _ln(sd, p + "output.LayerNorm", _lin(sd, p + "output.dense", ctx) + h)
    return out, kv


def layer(sd, p, h, self_mask, H, enc=None, enc_mask=None, past_kv=None, cross_cache=None):
    """models/med.py:333-383 with mode='multimodal' when enc is given."""
    a, kv = attention_block(sd, p + "attention.", h, self_mask, H, past_kv)
    if enc is not None:
        a, _ = attention_block(sd, p + "crossattention.", a, enc_mask, H, None, enc, cross_cache)
    inter = F.gelu(_lin(sd, p + "intermediate.dense", a))  # models/med.py:291-303, hidden_act 'gelu' (erf)
    out = _ln(sd, p + "output.LayerNorm", _lin(sd, p + "output.dense", inter) + a)  # :306-317
    return out, kv


def extended_mask(attention_mask, T, is_decoder):
    """models/med.py:609-668: (1 - m) * -10000, causal ∧ padding for the decoder."""
    B, S = attention_mask.shape  # S = past + T
    m = attention_mask.to(torch.float32)
    if is_decoder:
        ids = torch.arange(T)
        causal = (ids[None, None, :].repeat(B, T, 1) <= ids[None, :, None]).to(torch.float32)
        if S > T:
            causal = torch.cat([torch.ones(B, T, S - T), causal], dim=-1)
        ext = causal[:, None, :, :] * m[:, None, None, :]
    else:
        ext = m[:, None, None, :]
    return (1.0 - ext) * -10000.0


def bert_model(sd, p, ids, attention_mask, *, layers=12, H=12, enc=None, is_decoder=False, past=None, cross_cache=None):
    """models/med.py:670-807 (BertModel.forward).  Returns (hidden [B,T,C], new cache)."""
    B, T = ids.shape
    past_len = 0 if past is None else past[0][0].shape[2]
    if attention_mask is None:
        attention_mask = torch.ones(B, T + past_len)
    self_mask = extended_mask(attention_mask, T, is_decoder)
    enc_mask = None  # image_atts is all ones -> invert_attention_mask gives zeros (models/med.py:756-758)
    h = embeddings(sd, p + "embeddings.", ids, past_len)
    cache = []
    for i in range(layers):
        h, kv = layer(sd, f"{p}encoder.layer.{i}.", h, self_mask, H, enc, enc_mask, None if past is None else past[i],
                      cross_cache)
        cache.append(kv)
    return h, cache


def lm_head(sd, p, h):
    """models/med.py:501-545: dense, GELU, LayerNorm, decoder (+bias)."""
    t = _ln(sd, p + "predictions.transform.LayerNorm", F.gelu(_lin(sd, p + "predictions.transform.dense", h)))
    return F.linear(t, sd[p + "predictions.decoder.weight"], sd[p + "predictions.bias"])


def decoder_logits(sd, ids, enc, past=None, *, prefix="text_decoder.", layers=12, H=12, cross_cache=None):
    """One BertLMHeadModel.forward as HF generate() drives it (models/med.py:830-949):
    with a cache only the last token is fed.  Returns (last-position logits [B,V], cache)."""
    if past is not None:
        ids = ids[:, -1:]
    h, cache = bert_model(sd, prefix + "bert.", ids, None, layers=layers, H=H, enc=enc, is_decoder=True, past=past,
                          cross_cache=cross_cache)
    return lm_head(sd, prefix + "cls.", h[:, -1]), cache


def reorder_cache(past, beam_idx):
    """models/med.py:951-955."""
    return [tuple(t.index_select(0, beam_idx) for t in kv) for kv in past]


def itm_logits(sd, image_embeds, ids, attention_mask, *, prefix="", layers=12, H=12, cross_cache=None):
    """models/blip_itm.py:41-58 with match_head='itm', image_embeds given: [F,2] raw logits."""
    h, _ = bert_model(sd, prefix + "text_encoder.", ids, attention_mask, layers=layers, H=H, enc=image_embeds,
                      is_decoder=False, cross_cache=cross_cache)
    return F.linear(h[:, 0, :], sd[prefix + "itm_head.weight"], sd[prefix + "itm_head.bias"])


def filter_scores(itm_out):
    """run_video_CapFilt.py:114: softmax(dim=1)[:,1]."""
    return torch.softmax(itm_out, dim=1)[:, 1]
