"""Developer probe (CPU only, not a test; round 5): WHICH attention operands need more than 16 bits for the caption logits to stay
within 1e-3 ABSOLUTE of the fp32 reference at a trained model's logit scale (tests/common.trained_like_, max|logit| ~ 16-19)?

The fp32 oracle is run with a chosen precision injected into the Q / K / V / P operands of each attention SITE — the ViT's
self-attention, the decoder's self-attention, the decoder's cross-attention — and nothing else (every GEMM stays fp32, as the
error-compensated GEMMs of the parity precision mode nearly are):
    "f32"    untouched
    "f16"    rounded to f16 (what vidil_attention's kernels do)
    "split"  x = hi + lo, hi = f16(x), lo = f16(x - hi); a product a.b is a_hi.b_hi + a_lo.b_hi + a_hi.b_lo (the lo.lo term is
             dropped): the split-operand MFMA attention of round 5 (vidil_attention_f32 arith = 1)
    "hi8"    x = f16(x) + e4m3(x - f16(x)) scaled per row: three bytes per element
Printed: max |logit - fp32 logit| of the prompt pass and the same as a fraction of max|logit|.  The budget is 1e-3 / scale."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.nn.functional as F
from common import trained_like_, perturb_, synthetic_frames
from vidil_amd.blip import BLIP_Decoder
from vidil_amd.tokenizer import SyntheticBertTokenizer
from oracle import clip_ref, vit_ref, med_ref

torch.set_num_threads(8)
r16 = lambda t: t.half().float()


def parts(x, kind):
    """-> list of (tensor, order) parts whose sum represents x; order 0 = leading, 1 = correction."""
    if kind == "f32":
        return [(x, 0)]
    hi = r16(x)
    if kind == "f16":
        return [(hi, 0)]
    if kind == "split":
        return [(hi, 0), (r16(x - hi), 1)]
    if kind == "hi8":
        lo = x - hi
        sc = lo.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 224.0
        return [(hi, 0), ((lo / sc).to(torch.float8_e4m3fn).float() * sc, 1)]
    raise ValueError(kind)


def prod(a_parts, b_parts, f):
    """sum of f(a_i, b_j) over part pairs, dropping correction x correction."""
    out = None
    for a, oa in a_parts:
        for b, ob in b_parts:
            if oa + ob > 1:
                continue
            t = f(a, b)
            out = t if out is None else out + t
    return out


CFG = {}   # site -> dict(q=, k=, v=, p=)


def attn_core(site, q, k, v, add_mask=None):
    c = CFG.get(site, {})
    s = prod(parts(q, c.get("q", "f32")), parts(k, c.get("k", "f32")), lambda a, b: a @ b.transpose(-1, -2))
    if add_mask is not None:
        s = s + add_mask
    e = torch.exp(s - s.amax(-1, keepdim=True))
    l = e.sum(-1, keepdim=True)
    o = prod(parts(e, c.get("p", "f32")), parts(v, c.get("v", "f32")), lambda a, b: a @ b)
    return o / l


def vit_attention(sd, p, x, heads):
    B, N, C = x.shape
    qkv = F.linear(x, sd[p + "qkv.weight"], sd.get(p + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * ((C // heads) ** -0.5), qkv[1], qkv[2]
    y = attn_core("vit", q, k, v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"])


def med_self_attention(sd, p, h, add_mask, H, past_kv=None, kv_src=None, cross_cache=None):
    q = med_ref._heads(med_ref._lin(sd, p + "query", h), H) / math.sqrt(64)
    src = h if kv_src is None else kv_src
    k = med_ref._heads(med_ref._lin(sd, p + "key", src), H)
    v = med_ref._heads(med_ref._lin(sd, p + "value", src), H)
    if kv_src is None and past_kv is not None:
        k = torch.cat([past_kv[0], k], dim=2); v = torch.cat([past_kv[1], v], dim=2)
    ctx = attn_core("cross" if kv_src is not None else "dself", q, k, v, add_mask).permute(0, 2, 1, 3).contiguous()
    return ctx.view(ctx.shape[0], ctx.shape[1], -1), (k, v)


vit_ref.attention = vit_attention
med_ref.self_attention = med_self_attention
ALL = lambda kind: dict(q=kind, k=kind, v=kind, p=kind)
CONFIGS = [
    ("all f16 (vidil_attention)", dict(vit=ALL("f16"), dself=ALL("f16"), cross=ALL("f16"))),
    ("all split", dict(vit=ALL("split"), dself=ALL("split"), cross=ALL("split"))),
    ("only ViT f16", dict(vit=ALL("f16"))),
    ("only decoder self f16", dict(dself=ALL("f16"))),
    ("only cross f16", dict(cross=ALL("f16"))),
    ("cross: K V f16, Q P split; rest split", dict(vit=ALL("split"), dself=ALL("split"), cross=dict(q="split", k="f16", v="f16", p="split"))),
    ("cross: K f16 only", dict(cross=dict(k="f16"))),
    ("cross: V f16 only", dict(cross=dict(v="f16"))),
    ("cross: Q f16 only", dict(cross=dict(q="f16"))),
    ("cross: P f16 only", dict(cross=dict(p="f16"))),
    ("cross: K V hi8, Q P split; rest split", dict(vit=ALL("split"), dself=ALL("split"), cross=dict(q="split", k="hi8", v="hi8", p="split"))),
    ("cross: K hi8, V f16, Q P split; rest split", dict(vit=ALL("split"), dself=ALL("split"), cross=dict(q="split", k="hi8", v="f16", p="split"))),
    ("decoder self: K V f16, Q P split", dict(dself=dict(q="split", k="f16", v="f16", p="split"))),
    ("decoder self: V f16 only", dict(dself=dict(v="f16"))),
    ("decoder self: K f16 only", dict(dself=dict(k="f16"))),
    ("ViT: K V f16 only", dict(vit=dict(k="f16", v="f16"))),
    ("ViT: Q f16 only", dict(vit=dict(q="f16"))),
    ("ViT: P f16 only", dict(vit=dict(p="f16"))),
]
if __name__ == "__main__":
    for name, hs, tl in (("trained-like head x2", 2.0, True), ("random-init + perturb", 1.0, False)):
        torch.manual_seed(0)
        cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=SyntheticBertTokenizer()).eval()
        if tl:
            trained_like_(cap, 300, head_scale=hs, stream_shift=8.0)
        else:
            perturb_(cap, 100)
        sd = {k: v.clone() for k, v in cap.state_dict().items()}
        u8 = synthetic_frames(1, 2, first_video=21)[0]
        ids = cap.prompt_ids(2, "cpu").long().repeat_interleave(3, 0)

        def run():
            with torch.no_grad():
                y = vit_ref.vit_forward(sd, clip_ref.preprocess_u8(u8))
                lg, _ = med_ref.decoder_logits(sd, ids, y.repeat_interleave(3, 0))
            return y, lg

        CFG.clear()
        y0, l0 = run()
        scale = l0.abs().max().item()
        print(f"== {name}: max|logit| {scale:.1f}; budget 1e-3 absolute = {1e-3 / scale:.2e} of the scale")
        for label, cfg in CONFIGS:
            CFG.clear(); CFG.update(cfg)
            y, lg = run()
            dy, dl = (y - y0).abs().max().item(), (lg - l0).abs().max().item()
            print(f"  {label:55s} ViT out {dy:.2e}   logits {dl:.2e} = {dl / scale:.2e} of the scale")
