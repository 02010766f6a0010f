"""CLIP towers on the HIP kernels — stand-in for the HuggingFace ``CLIPModel`` /
``CLIPProcessor`` objects the reference drives in run_visual_tokenization.py
(:83-96 text embeddings, :135-143 image embeddings, :347-350 construction), with the same call signatures:
``CLIPModel.from_pretrained(name)``, ``CLIPProcessor.from_pretrained(name)``, ``processor(text=..., images=...,
return_tensors='pt', padding=True).to(device)``, ``model(**inputs)``.

Same parameter names as HF's ``CLIPModel.state_dict()`` (so ``openai/clip-vit-*``
weights load with ``load_state_dict``), same outputs (``.image_embeds`` /
``.text_embeds``, unit-norm), plus ``encode_image`` / ``encode_text``.  Unlike the
reference's call pattern, ``model(**inputs)`` runs only the tower whose inputs are
given (the reference feeds a dummy image to text batches and 'hello world' to image
batches, run_visual_tokenization.py:90-91,138-141).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import kernels as K
from .packing import FP8, PackedCache, fold_layernorm, parity_attention_arith, parity_attention_f32, require_cuda, v32, w3, w3_patch, w8, w16, w16_patch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class CLIPVisionConfig:
    def __init__(self, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 image_size=224, patch_size=32, layer_norm_eps=1e-5, **_):
        self.__dict__.update(locals()); self.__dict__.pop("self"); self.__dict__.pop("_", None)


class CLIPTextConfig:
    def __init__(self, vocab_size=49408, hidden_size=512, intermediate_size=2048, num_hidden_layers=12,
                 num_attention_heads=8, max_position_embeddings=77, layer_norm_eps=1e-5, eos_token_id=49407,
                 bos_token_id=49406, pad_token_id=1, **_):
        self.__dict__.update(locals()); self.__dict__.pop("self"); self.__dict__.pop("_", None)


class CLIPConfig:
    """Defaults = openai/clip-vit-base-patch32; ``CLIPConfig.vit_l14()`` = the reference YAMLs' default model."""

    def __init__(self, vision=None, text=None, projection_dim=512):
        self.vision_config = vision or CLIPVisionConfig()
        self.text_config = text or CLIPTextConfig()
        self.projection_dim = projection_dim

    @classmethod
    def vit_l14(cls):
        return cls(CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                    num_attention_heads=16, patch_size=14),
                   CLIPTextConfig(hidden_size=768, intermediate_size=3072, num_attention_heads=12), 768)


# ---- parameter holders with HF's names --------------------------------------------------
class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj = nn.Linear(d, d)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class _MLP(nn.Module):
    def __init__(self, d, inter):
        super().__init__()
        self.fc1 = nn.Linear(d, inter)
        self.fc2 = nn.Linear(inter, d)


class _Layer(nn.Module):
    def __init__(self, d, inter, eps):
        super().__init__()
        self.self_attn = _Attn(d)
        self.layer_norm1 = nn.LayerNorm(d, eps=eps)
        self.mlp = _MLP(d, inter)
        self.layer_norm2 = nn.LayerNorm(d, eps=eps)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(cfg.hidden_size, cfg.intermediate_size, cfg.layer_norm_eps)
                                     for _ in range(cfg.num_hidden_layers)])


class _VisionEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(cfg.hidden_size))
        self.patch_embedding = nn.Conv2d(3, cfg.hidden_size, kernel_size=cfg.patch_size, stride=cfg.patch_size, bias=False)
        n = (cfg.image_size // cfg.patch_size) ** 2 + 1
        self.position_embedding = nn.Embedding(n, cfg.hidden_size)


class _VisionModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _VisionEmbeddings(cfg)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)   # (sic) HF's spelling
        self.encoder = _Encoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _TextEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)


class _TextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _TextEmbeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


def _pack_layers(encoder, c, fuse=False, fp8=False, parity=False):
    """fuse: LayerNorm folded into the QKV / fc1 GEMMs (layer_norm2 everywhere, layer_norm1 from layer 1 on: layer
    0's input is written by a stand-alone LayerNorm / embedding kernel, not by a residual GEMM).
    parity: [W_hi | W_hi | W_lo] operands of the error-compensated GEMMs (packing.set_parity_mode)."""
    out = []
    for i, l in enumerate(encoder.layers):
        a = l.self_attn
        extra = {}
        if parity:
            extra["qkv_w3"] = w3(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, dtype=c)
            extra["o_w3"] = w3(a.out_proj.weight, dtype=c)
            extra["fc1_w3"] = w3(l.mlp.fc1.weight, dtype=c)
            extra["fc2_w3"] = w3(l.mlp.fc2.weight, dtype=c)
        elif fp8:     # fp8 tower mode (see vit.VisionTransformer._run_blocks_fp8)
            extra["qkv_w8"], extra["qkv_s"] = w8(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight)
            extra["o_w8"], extra["o_s"] = w8(a.out_proj.weight)
            extra["fc1_w8"], extra["fc1_s"] = w8(l.mlp.fc1.weight)
            extra["fc2_w8"], extra["fc2_s"] = w8(l.mlp.fc2.weight)
        elif fuse:
            extra["fc1_f"] = fold_layernorm(l.mlp.fc1.weight, l.mlp.fc1.bias, l.layer_norm2.weight, l.layer_norm2.bias, c)
            if i > 0:
                qkv_w = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], dim=0)
                qkv_b = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], dim=0)
                extra["qkv_f"] = fold_layernorm(qkv_w, qkv_b, l.layer_norm1.weight, l.layer_norm1.bias, c)
        out.append(dict(extra, 
            n1g=v32(l.layer_norm1.weight), n1b=v32(l.layer_norm1.bias),
            qkv_w=w16(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, dtype=c),
            qkv_b=v32(a.q_proj.bias, a.k_proj.bias, a.v_proj.bias),
            o_w=w16(a.out_proj.weight, dtype=c), o_b=v32(a.out_proj.bias),
            n2g=v32(l.layer_norm2.weight), n2b=v32(l.layer_norm2.bias),
            fc1_w=w16(l.mlp.fc1.weight, dtype=c), fc1_b=v32(l.mlp.fc1.bias),
            fc2_w=w16(l.mlp.fc2.weight, dtype=c), fc2_b=v32(l.mlp.fc2.bias)))
    return out


def _run_layers(layers, x, B, T, H, eps, *, causal=False, kv_len=None, f32_attn=True, arith=0, cls_last=False):
    """Pre-LN CLIP encoder layers on the f32 residual stream x [B*T, D] (in place).
    cls_last (the vision tower in the parity precision mode with the f32-row attention kinds, round 6): the caller reads token 0
    of every image only (pooled output = post_layernorm(CLS), HF CLIPVisionTransformer), so the LAST layer computes K | V for all
    rows and everything else — the query, the attention output, out-proj, LayerNorm 2, fc1, fc2 — for the B class-token rows alone
    (10/12 of that layer's GEMM rows are not computed: ~7 % of the tower); returns the f32 CLS rows [B, D] instead of x.
    Per class-token row the arithmetic is the full layer's (its attention in plain f32 arithmetic instead of the split form)."""
    dev = x.device
    M, D = x.shape
    cdt = layers[0]["qkv_w"].dtype
    # more than 32 rows -> LDS-staged attention, which takes V row-major (NP = 0: plain 16-B stores from the QKV GEMM);
    # short text batches go through the direct kernels, which read V^T fragments straight from memory
    NP = 0 if T > 32 else (T + 15) // 16 * 16
    xn = torch.empty((M, D), dtype=cdt, device=dev)
    q = torch.empty((B, H, T, 64), dtype=cdt, device=dev)
    k = torch.empty((B, H, T, 64), dtype=cdt, device=dev)
    vt = torch.empty((B, H, T, 64) if NP == 0 else (B, H, 64, NP), dtype=cdt, device=dev)
    o = torch.empty((M, D), dtype=cdt, device=dev)
    hid = torch.empty((M, layers[0]["fc1_w"].shape[0]), dtype=cdt, device=dev)
    heads = dict(q=q, k=k, vt=vt, T=T, H=H, part0=0, t_off=0, Tq_cap=T, Tk_cap=T, NP=NP, q_scale=0.125)
    n = len(layers)
    if "qkv_w3" in layers[0]:
        # Parity precision mode (round 4; as vit.VisionTransformer._run_blocks_parity): every GEMM on error-compensated
        # operands — LayerNorm and attention write [hi | lo | hi] rows (VIDIL_DT_SPLIT3), the quick-GELU output goes through
        # the GEMM's own split3 epilogue, weights are [W_hi | W_hi | W_lo], K tripled.  Attention: vidil_attention_f32 on the f32
        # Q | K | V rows (split-operand MFMA by default, or f32 arithmetic), or — kind "16" — the 16-bit kernels.
        Dh = layers[0]["fc1_w"].shape[0]
        a3 = torch.empty((M, 3 * D), dtype=cdt, device=dev)
        o3 = torch.empty((M, 3 * D), dtype=cdt, device=dev)
        hid3 = torch.empty((M, 3 * Dh), dtype=cdt, device=dev)
        # (planes hi | lo only where EVERY consumer of the rows takes the K-loop form — asked per call, vit.py has the reasoning;
        #  the consumers state a_planes so that a launch that would read an unwritten plane fails instead)
        qkv32 = torch.empty((M, 3 * D), dtype=torch.float32, device=dev) if f32_attn else None
        planes, l0 = 3, layers[0]
        if K.split_k_in_loop():
            qkv_kw = dict(out=qkv32) if f32_attn else dict(heads=heads)
            if (K.split_k_serves(a3, l0["qkv_w3"], l0["qkv_b"], **qkv_kw) and K.split_k_serves(o3, l0["o_w3"], l0["o_b"], out=x, resid=x)
                    and K.split_k_serves(a3, l0["fc1_w3"], l0["fc1_b"], split3_out=hid3, act=K.ACT_QUICK_GELU)
                    and K.split_k_serves(hid3, l0["fc2_w3"], l0["fc2_b"], out=x, resid=x)):
                planes = 2
        import os
        if planes == 2 and os.environ.get("VIDIL_POISON_SPLIT3") == "1":       # (developer: NaNs in the unwritten third planes —
            for _b in (a3, o3, hid3,):                               #  any consumer that reads one shows up at once)
                _b[:, 2 * (_b.shape[1] // 3):] = float("nan")
        cls_last = cls_last and f32_attn and not causal and kv_len is None and T > 1
        for li, l in enumerate(layers):
            K.layernorm(x, l["n1g"], l["n1b"], eps, out16=a3, split3=True, planes=planes)
            if cls_last and li == n - 1:
                # ---- last layer, class-token rows only (K | V still for every token)
                kv32 = qkv32.view(-1)[:M * 2 * D].view(M, 2 * D)               # (the layers' scratch, re-shaped: [M, 2D] keys | values)
                K.gemm(a3, l["qkv_w3"][D:], l["qkv_b"][D:], out=kv32, split_k=True, a_planes=planes)
                a3c = a3.view(B, T, 3 * D)[:, 0].contiguous()                  # [B, 3D] operand rows of the class tokens
                if planes == 2:
                    a3c[:, 2 * D:] = a3c[:, :D]                                # (three valid planes: the small launches below may be plain)
                q32c = K.gemm(a3c, l["qkv_w3"][:D], l["qkv_b"][:D], out_dtype=torch.float32, split_k=True)
                o3c = torch.empty((B, 3 * D), dtype=cdt, device=dev)
                K.attention_f32(q32c, kv32[:, :D], kv32[:, D:], o3c, Bq=B, H=H, Nq=1, Nk=T, kv_rows=T, arith=0, planes=3)
                xc = x.view(B, T, D)[:, 0].contiguous()                        # [B, D] f32 residual rows of the class tokens
                K.gemm(o3c, l["o_w3"], l["o_b"], out=xc, resid=xc, split_k=True)
                K.layernorm(xc, l["n2g"], l["n2b"], eps, out16=a3c, split3=True, planes=3)
                h3c = K.gemm(a3c, l["fc1_w3"], l["fc1_b"], split3_out=torch.empty((B, 3 * Dh), dtype=cdt, device=dev), act=K.ACT_QUICK_GELU,
                             split_k=True, split3_planes=3)
                K.gemm(h3c, l["fc2_w3"], l["fc2_b"], out=xc, resid=xc, split_k=True)
                return xc
            if f32_attn:    # (Q | K | V stay f32 and row-major: vidil_attention_f32 reads them in place)
                K.gemm(a3, l["qkv_w3"], l["qkv_b"], out=qkv32, split_k=True, a_planes=planes)
                K.attention_f32(qkv32[:, :D], qkv32[:, D:2 * D], qkv32[:, 2 * D:], o3, Bq=B, H=H, Nq=T, Nk=T, causal=causal, kv_len=kv_len,
                                arith=arith, planes=planes)
            else:
                K.gemm(a3, l["qkv_w3"], l["qkv_b"], heads=heads, split_k=True, a_planes=planes)
                K.attention(q, k, vt, o3, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=NP, causal=causal, kv_len=kv_len, split3=True)
            K.gemm(o3, l["o_w3"], l["o_b"], out=x, resid=x, split_k=True, a_planes=planes if f32_attn else 3)
            K.layernorm(x, l["n2g"], l["n2b"], eps, out16=a3, split3=True, planes=planes)
            K.gemm(a3, l["fc1_w3"], l["fc1_b"], split3_out=hid3, act=K.ACT_QUICK_GELU, split_k=True, split3_planes=planes, a_planes=planes)
            K.gemm(hid3, l["fc2_w3"], l["fc2_b"], out=x, resid=x, split_k=True, a_planes=planes)
        return x
    stats = torch.empty((M, D // 64, 2), dtype=torch.float32, device=dev) if "fc1_f" in layers[0] else None
    if "qkv_w8" in layers[0] and T > 32:
        xn8 = torch.empty((M, D), dtype=FP8, device=dev)
        o8 = torch.empty((M, D), dtype=FP8, device=dev)
        hid8 = torch.empty((M, layers[0]["fc1_w8"].shape[0]), dtype=FP8, device=dev)
        for l in layers:
            K.layernorm(x, l["n1g"], l["n1b"], eps, out16=xn8)
            K.gemm(xn8, l["qkv_w8"], l["qkv_b"], heads=heads, w_scale=l["qkv_s"])
            K.attention(q, k, vt, o8, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=NP, causal=causal, kv_len=kv_len)
            K.gemm(o8, l["o_w8"], l["o_b"], out=x, resid=x, w_scale=l["o_s"], dtype16=cdt)
            K.layernorm(x, l["n2g"], l["n2b"], eps, out16=xn8)
            K.gemm(xn8, l["fc1_w8"], l["fc1_b"], out=hid8, act=K.ACT_QUICK_GELU, w_scale=l["fc1_s"], dtype16=cdt)
            K.gemm(hid8, l["fc2_w8"], l["fc2_b"], out=x, resid=x, w_scale=l["fc2_s"], dtype16=cdt)
        return x
    for i, l in enumerate(layers):
        fused = "fc1_f" in l      # LayerNorm folded into the consuming GEMMs (see vit.VisionTransformer.run_blocks)
        if fused and i > 0:
            w_, b_, cs = l["qkv_f"]
            K.gemm(xn, w_, b_, heads=heads, ln=(cs, eps, stats))
        else:
            K.layernorm(x, l["n1g"], l["n1b"], eps, out16=xn)
            K.gemm(xn, l["qkv_w"], l["qkv_b"], heads=heads)
        K.attention(q, k, vt, o, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=NP, causal=causal, kv_len=kv_len)
        if fused:
            K.gemm(o, l["o_w"], l["o_b"], out=x, resid=x, out16=xn, ln_stats_out=stats)
            w_, b_, cs = l["fc1_f"]
            K.gemm(xn, w_, b_, out=hid, act=K.ACT_QUICK_GELU, ln=(cs, eps, stats))
            K.gemm(hid, l["fc2_w"], l["fc2_b"], out=x, resid=x, out16=xn if i + 1 < n else None,
                   ln_stats_out=stats if i + 1 < n else None)
        else:
            K.gemm(o, l["o_w"], l["o_b"], out=x, resid=x)
            K.layernorm(x, l["n2g"], l["n2b"], eps, out16=xn)
            K.gemm(xn, l["fc1_w"], l["fc1_b"], out=hid, act=K.ACT_QUICK_GELU)
            K.gemm(hid, l["fc2_w"], l["fc2_b"], out=x, resid=x)
    return x


class CLIPModel(PackedCache, nn.Module):
    def __init__(self, config: CLIPConfig = None):
        super().__init__()
        self.config = config or CLIPConfig()
        vc, tc = self.config.vision_config, self.config.text_config
        for c in (vc, tc):
            if c.hidden_size // c.num_attention_heads != 64:
                raise ValueError("vidil_amd CLIP kernels are built for head_dim 64")
        self.text_model = _TextModel(tc)
        self.vision_model = _VisionModel(vc)
        self.visual_projection = nn.Linear(vc.hidden_size, self.config.projection_dim, bias=False)
        self.text_projection = nn.Linear(tc.hidden_size, self.config.projection_dim, bias=False)
        self.logit_scale = nn.Parameter(torch.tensor(2.6592))
        import os
        self.fuse_layernorm = os.environ.get("VIDIL_FUSE_LN", "1") != "0"   # vision tower only (the text tower runs once per ontology)
        # (parity precision mode: the vision tower's last layer on the class-token rows only — _run_layers(cls_last=True); A/B switch)
        self.cls_only_last_layer = os.environ.get("VIDIL_CLIP_CLS_LAST", "1") != "0"
        self.apply(self._init)

    @classmethod
    def from_pretrained(cls, name_or_path, state_dict=None, **kw):
        """``CLIPModel.from_pretrained(name)`` of the reference (run_visual_tokenization.py:347-348).  ``name_or_path``: a
        directory holding ``config.json`` + ``model.safetensors`` / ``pytorch_model.bin`` (the layout ``save_pretrained``
        and the hub use), or a hub id, resolved through the local HF cache / the hub with ``huggingface_hub``.
        ``state_dict``: load these tensors instead of the files (config still from ``name_or_path`` if it is a
        directory, else the ViT-B/32 defaults)."""
        import json
        import os

        path = str(name_or_path)
        if not os.path.isdir(path) and state_dict is None:
            from huggingface_hub import snapshot_download

            path = snapshot_download(path, allow_patterns=["config.json", "preprocessor_config.json", "model.safetensors",
                                                           "pytorch_model.bin"], **kw)
        cfg = None
        cj = os.path.join(path, "config.json")
        if os.path.isfile(cj):
            with open(cj) as f:
                c = json.load(f)
            cfg = CLIPConfig(CLIPVisionConfig(**c.get("vision_config", {})), CLIPTextConfig(**c.get("text_config", {})),
                             c.get("projection_dim", 512))
        model = cls(cfg)
        if state_dict is None:
            st = os.path.join(path, "model.safetensors")
            if os.path.isfile(st):
                from safetensors.torch import load_file

                state_dict = load_file(st)
            else:
                state_dict = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        state_dict = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}    # buffers of older HF versions
        msg = model.load_state_dict(state_dict, strict=False)
        if msg.missing_keys:
            raise RuntimeError(f"CLIPModel.from_pretrained: checkpoint lacks {msg.missing_keys[:5]} ...")
        if msg.unexpected_keys:
            raise RuntimeError(f"CLIPModel.from_pretrained: unexpected keys {msg.unexpected_keys[:5]} ...")
        return model.eval()

    @staticmethod
    def _init(m):
        if isinstance(m, (nn.Linear, nn.Embedding)):
            nn.init.normal_(m.weight, std=0.02)
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)

    def pack_flags(self):
        return (self.fuse_layernorm, self.fp8, self.parity)

    def _pack(self):
        vm, tm = self.vision_model, self.text_model
        D = self.config.vision_config.hidden_size
        c = self.cdt
        par = self.parity
        extra = {}
        if par:
            # parity precision mode (round 4: packing.set_parity_mode(True, clip) / $VIDIL_PARITY): both towers and both
            # projections on error-compensated operands, so that the embeddings the exact-f32 ontology scan reads carry the
            # fp32 reference's values to ~1e-6 and its top-k indices can be compared END TO END (tests/test_parity_mode_gpu.py)
            extra = dict(pe_w3=w3_patch(vm.embeddings.patch_embedding.weight, c), vproj3=w3(self.visual_projection.weight, dtype=c),
                         tproj3=w3(self.text_projection.weight, dtype=c))
        return dict(extra, parity=par,
            pe_w=w16_patch(vm.embeddings.patch_embedding.weight, c),
            cls=v32(vm.embeddings.class_embedding), pos=v32(vm.embeddings.position_embedding.weight).view(-1, D),
            pre_g=v32(vm.pre_layrnorm.weight), pre_b=v32(vm.pre_layrnorm.bias),
            post_g=v32(vm.post_layernorm.weight), post_b=v32(vm.post_layernorm.bias),
            vproj=w16(self.visual_projection.weight, dtype=c),
            vlayers=_pack_layers(vm.encoder, c, self.fuse_layernorm and not par, self.fp8 and not par, par),
            tok=v32(tm.embeddings.token_embedding.weight).view(self.config.text_config.vocab_size, -1),
            tpos=v32(tm.embeddings.position_embedding.weight).view(self.config.text_config.max_position_embeddings, -1),
            fin_g=v32(tm.final_layer_norm.weight), fin_b=v32(tm.final_layer_norm.bias),
            tproj=w16(self.text_projection.weight, dtype=c), tlayers=_pack_layers(tm.encoder, c, parity=par))

    # ------------------------------------------------------------------ vision tower
    def _vision_from_patches(self, patches16, B, pooled=False):
        p = self.packed()
        vc = self.config.vision_config
        D, H = vc.hidden_size, vc.num_attention_heads
        P = (vc.image_size // vc.patch_size) ** 2
        T = P + 1
        dev = patches16.device
        cdt = patches16.dtype
        par = p["parity"]          # (then patches16 holds [hi | lo | hi] rows: see the callers)
        x = torch.empty((B * T, D), dtype=torch.float32, device=dev)
        K.gemm(patches16, p["pe_w3"] if par else p["pe_w"], None, patch=dict(out=x, pos=p["pos"], tpi=P), split_k=par)
        K.set_cls_row(x, p["cls"], p["pos"], B, T, D)
        K.layernorm(x, p["pre_g"], p["pre_b"], vc.layer_norm_eps, out32=x)
        cls_last = par and self.cls_only_last_layer
        xo = _run_layers(p["vlayers"], x, B, T, H, vc.layer_norm_eps, f32_attn=parity_attention_f32(self), arith=parity_attention_arith(self),
                         cls_last=cls_last)
        cls_rows = xo.shape[0] == B and T > 1          # (the layers handed back the class-token rows [B, D] instead of the stream)
        pooled16 = torch.empty((B, (3 if par else 1) * D), dtype=cdt, device=dev)
        pooled32 = torch.empty((B, D), dtype=torch.float32, device=dev) if pooled else None
        K.layernorm(xo if cls_rows else x, p["post_g"], p["post_b"], vc.layer_norm_eps, M=B, D=D, x_stride=D if cls_rows else T * D,
                    out16=pooled16, out32=pooled32, split3=par)
        if pooled:
            return pooled32
        emb = K.gemm(pooled16, p["vproj3"] if par else p["vproj"], None, out_dtype=torch.float32, split_k=par)
        return K.l2_normalize_rows(emb)

    @torch.no_grad()
    def encode_image(self, pixel_values):
        """pixel_values f32 [F,3,S,S] (already normalised) -> unit-norm f32 [F,P]."""
        require_cuda(pixel_values, "CLIPModel.encode_image")
        ps = self.config.vision_config.patch_size
        patches = K.patchify_f32(pixel_values.contiguous().float(), ps, dtype=self.cdt, split3=self.parity)
        return self._vision_from_patches(patches, pixel_values.shape[0])

    @torch.no_grad()
    def encode_image_u8(self, frames_u8):
        """uint8 [F,S,S,3] frames already at the model resolution; fused /255 + CLIP normalisation."""
        require_cuda(frames_u8, "CLIPModel.encode_image_u8")
        ps = self.config.vision_config.patch_size
        patches = K.patchify_u8(frames_u8.contiguous(), ps, CLIP_MEAN, CLIP_STD, dtype=self.cdt, split3=self.parity)
        return self._vision_from_patches(patches, frames_u8.shape[0])

    @torch.no_grad()
    def pooled_image_u8(self, frames_u8):
        """uint8 [F,S,S,3] -> f32 [F,D]: HF's vision ``pooler_output`` (post_layernorm of the class token, before
        visual_projection) — what the reference's 'clip-kmeans' frame selection clusters
        (data/video_pretrain_dataset.py:199-202)."""
        require_cuda(frames_u8, "CLIPModel.pooled_image_u8")
        ps = self.config.vision_config.patch_size
        patches = K.patchify_u8(frames_u8.contiguous(), ps, CLIP_MEAN, CLIP_STD, dtype=self.cdt, split3=self.parity)
        return self._vision_from_patches(patches, frames_u8.shape[0], pooled=True)

    # ------------------------------------------------------------------ text tower
    @torch.no_grad()
    def encode_text(self, input_ids, attention_mask=None):
        """ids [N,L] (int) -> unit-norm f32 [N,P]; pooled at the first EOS (HF >= 4.31 rule; argmax rule when
        eos_token_id == 2).  Right padding is harmless under the causal mask, as in HF."""
        require_cuda(input_ids, "CLIPModel.encode_text")
        p = self.packed()
        tc = self.config.text_config
        D, H = tc.hidden_size, tc.num_attention_heads
        N, L = input_ids.shape
        dev = input_ids.device
        ids32 = input_ids.to(torch.int32).contiguous()
        cdt = p["tproj"].dtype
        x = torch.empty((N * L, D), dtype=torch.float32, device=dev)
        K.embed_tokens(ids32.view(-1), p["tok"], p["tpos"], x, T=L, pos_off=0)
        kv_len = None
        if attention_mask is not None:
            kv_len = attention_mask.to(dev).sum(dim=1).to(torch.int32).contiguous()
        _run_layers(p["tlayers"], x, N, L, H, tc.layer_norm_eps, causal=True, kv_len=kv_len, f32_attn=parity_attention_f32(self),
                    arith=parity_attention_arith(self))
        if tc.eos_token_id == 2:
            pos = ids32.argmax(dim=-1)
        else:
            pos = (ids32 == tc.eos_token_id).to(torch.int32).argmax(dim=-1)
        rows = (torch.arange(N, device=dev) * L + pos).to(torch.int32)
        sel = K.gather_rows(x, rows)
        par = p["parity"]
        pooled16 = torch.empty((N, (3 if par else 1) * D), dtype=cdt, device=dev)
        K.layernorm(sel, p["fin_g"], p["fin_b"], tc.layer_norm_eps, out16=pooled16, split3=par)
        emb = K.gemm(pooled16, p["tproj3"] if par else p["tproj"], None, out_dtype=torch.float32, split_k=par)
        return K.l2_normalize_rows(emb)

    def forward(self, input_ids=None, pixel_values=None, attention_mask=None, **_):
        out = SimpleNamespace(image_embeds=None, text_embeds=None)
        if pixel_values is not None:      # uint8 [F,S,S,3] from CLIPProcessor (rescale + normalise fused), or HF-style f32 [F,3,S,S]
            out.image_embeds = (self.encode_image_u8(pixel_values) if pixel_values.dtype == torch.uint8
                                else self.encode_image(pixel_values))
        if input_ids is not None:
            out.text_embeds = self.encode_text(input_ids, attention_mask)
        return out


class CLIPProcessor:
    """Look-alike of HF ``CLIPProcessor`` as the reference drives it (run_visual_tokenization.py:90-93,138-142,347-350):

        processor = CLIPProcessor.from_pretrained(name)
        inputs = processor(text=[...], images=[PIL or HWC uint8 ...], return_tensors="pt", padding=True).to(device)
        out = model(**inputs)            # .image_embeds / .text_embeds

    Text goes through the CLIP BPE tokenizer (HF ``CLIPTokenizer`` loaded from ``name``, or any injected callable with
    the same call signature — the vocabulary is a download).  Images of ANY size are uploaded as uint8, resized on the
    GPU exactly as ``CLIPImageProcessor`` does with Pillow (shortest edge -> S bicubic, centre crop; bit-exact, see
    preprocess.clip_frames) and handed over as ``pixel_values`` = uint8 [F,S,S,3]; rescale (1/255) and mean/std
    normalisation are fused into the model's patch-extraction kernel (``CLIPModel.forward`` dispatches on the dtype), so
    no f32 image is ever materialised.  There is no CPU fallback: without a GPU the image side raises."""

    def __init__(self, tokenizer=None, image_size=224, device=None):
        self.tokenizer = tokenizer
        self.image_size = image_size
        self.device = device

    @classmethod
    def from_pretrained(cls, name_or_path, tokenizer=None, device=None, **kw):
        import json
        import os

        size = 224
        cfg = os.path.join(str(name_or_path), "preprocessor_config.json")
        if os.path.isfile(cfg):
            with open(cfg) as f:
                pc = json.load(f)
            sz = pc.get("crop_size", pc.get("size", 224))
            size = sz if isinstance(sz, int) else sz.get("height", sz.get("shortest_edge", 224))
        if tokenizer is None:
            from transformers import CLIPTokenizer

            tokenizer = CLIPTokenizer.from_pretrained(name_or_path, **kw)   # raises when the vocabulary is unavailable
        return cls(tokenizer=tokenizer, image_size=size, device=device)

    def _frames(self, images):
        import numpy as np

        from .preprocess import clip_frames

        dev = torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        if dev is None or dev.type != "cuda":
            raise K.VidilHipError("CLIPProcessor: the image side runs on the GPU (no CPU fallback); no HIP device visible")
        frames = list(images) if isinstance(images, (list, tuple)) else [images]
        arrs = []
        for f in frames:
            if torch.is_tensor(f):
                a = f
            else:
                if hasattr(f, "convert"):                       # PIL image: convert RGB as CLIPImageProcessor does
                    f = f.convert("RGB")
                a = torch.from_numpy(np.array(f))            # (a copy: PIL hands out read-only buffers)
            if a.dtype != torch.uint8 or a.dim() != 3 or a.shape[-1] != 3:
                raise K.VidilHipError(f"CLIPProcessor: images must be HWC uint8 RGB, got {a.dtype} {tuple(a.shape)}")
            arrs.append(a)
        out = [None] * len(arrs)
        by_shape = {}
        for i, a in enumerate(arrs):
            by_shape.setdefault(tuple(a.shape), []).append(i)
        for shape, idx in by_shape.items():                     # one batched resize per frame geometry
            batch = torch.stack([arrs[i] for i in idx]).to(dev)
            res = clip_frames(batch, self.image_size)
            for j, i in enumerate(idx):
                out[i] = res[j]
        return torch.stack(out)

    def __call__(self, text=None, images=None, return_tensors="pt", padding=True, truncation=True, **kw):
        from .tokenizer import Encoding

        if return_tensors != "pt":
            raise ValueError("CLIPProcessor: only return_tensors='pt' is supported")
        enc = Encoding()
        if text is not None:
            if self.tokenizer is None:
                raise K.VidilHipError("CLIPProcessor: no tokenizer (from_pretrained could not load one and none was injected)")
            t = self.tokenizer([text] if isinstance(text, str) else list(text), return_tensors="pt", padding=padding,
                               truncation=truncation, **kw)
            enc["input_ids"] = t["input_ids"]
            if "attention_mask" in t:
                enc["attention_mask"] = t["attention_mask"]
        if images is not None:
            enc["pixel_values"] = self._frames(images)
        return enc
