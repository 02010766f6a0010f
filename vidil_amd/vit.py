"""BLIP vision transformer on the HIP kernels.

Mirror of the reference ``VisionTransformer`` (models/vit.py:113-194): same
constructor arguments, same parameter names (so BLIP ``.pth`` checkpoints load
unchanged: ``patch_embed.proj.weight``, ``cls_token``, ``pos_embed``,
``blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}``, ``norm``), same
``forward(x[B,3,S,S]) -> [B,1+P,width]`` fp32 contract.  The arithmetic is:

    patchify (f32->T16 im2col) -> GEMM(+bias+pos, row remap) -> 12 x [
        [LN+]QKV GEMM (per-head Q/K/V scatter, q pre-scaled) -> softmax attention ->
        proj GEMM (+residual, in place, + T16 copy of the stream) ->
        [LN+]fc1 GEMM (+erf-GELU) -> fc2 GEMM (+residual, + T16 copy) ] -> LN

where "[LN+]" is the block's LayerNorm folded into the GEMM (statistics from the A fragments the GEMM streams,
normalisation applied to the accumulators; ``fuse_layernorm``) — only block 0's norm1 and the final norm run as
stand-alone LayerNorm kernels —

with the residual stream, LayerNorm statistics and softmax in f32 and the MFMA
operands in the model's compute dtype (f16 or bf16, packing.set_compute_dtype).
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from . import kernels as K
from .packing import FP8, PackedCache, fold_layernorm, require_cuda, v32, w3, w3_patch, w8, w16, w16_patch, parity_attention_arith, parity_attention_f32, parity_attention_kind


class PatchEmbed(nn.Module):
    """Parameter holder matching timm's PatchEmbed (``proj`` = Conv2d k=s=patch)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class VisionTransformer(PackedCache, nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=None,
                 use_grad_checkpointing=False, ckpt_layer=0):
        super().__init__()
        if embed_dim // num_heads != 64:
            raise ValueError("vidil_amd ViT kernels are built for head_dim 64")
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.ln_eps = 1e-6  # models/vit.py:142
        # LayerNorm folded into the QKV / fc1 GEMMs (vidil_gemm_args.ln_fold): on unless VIDIL_FUSE_LN=0
        import os
        self.fuse_layernorm = os.environ.get("VIDIL_FUSE_LN", "1") != "0"
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=self.ln_eps)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        # init rules of models/vit.py:163-174
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ packing
    def pack_flags(self):
        return (self.fuse_layernorm, self.fp8, self.parity, self.parity_last_blocks, parity_attention_kind(self))

    @property
    def parity_last_blocks(self):
        """Parity precision mode, MIXED form (round 4): only the last k blocks run on error-compensated operands, the
        first depth - k on plain 16-bit operands (unfused LayerNorm kernels + plain GEMMs: both forms meet at the f32
        residual stream).  None = all blocks (the default).  Rounding errors of the tower accumulate like a random walk over
        its 48 GEMMs, so compensating the last k of 12 blocks removes ~k/12 of the tower's error variance at ~k/12 of the
        3x cost — tests/probes/probe_parity_mix.py measures caption-logit error and time for every k."""
        return self.__dict__.get("_parity_last_blocks")

    def set_parity_last_blocks(self, k):
        if k is not None and not (0 <= int(k) <= len(self.blocks)):
            raise ValueError(f"parity_last_blocks must be in 0..{len(self.blocks)}")
        self.__dict__["_parity_last_blocks"] = None if k is None else int(k)

    def _pack(self):
        D = self.embed_dim
        pe = self.patch_embed.proj
        c = self.cdt
        p = dict(
            pe_w=w16_patch(pe.weight, c), pe_b=v32(pe.bias),
            cls=v32(self.cls_token), pos=v32(self.pos_embed).view(-1, D),
            norm_g=v32(self.norm.weight), norm_b=v32(self.norm.bias), blocks=[])
        for i, b in enumerate(self.blocks):
            d = dict(
                n1g=v32(b.norm1.weight), n1b=v32(b.norm1.bias),
                qkv_w=w16(b.attn.qkv.weight, dtype=c), qkv_b=v32(b.attn.qkv.bias),
                proj_w=w16(b.attn.proj.weight, dtype=c), proj_b=v32(b.attn.proj.bias),
                n2g=v32(b.norm2.weight), n2b=v32(b.norm2.bias),
                fc1_w=w16(b.mlp.fc1.weight, dtype=c), fc1_b=v32(b.mlp.fc1.bias),
                fc2_w=w16(b.mlp.fc2.weight, dtype=c), fc2_b=v32(b.mlp.fc2.bias))
            if self.parity:
                # parity precision mode (packing.set_parity_mode): [W_hi | W_hi | W_lo] against [x_hi | x_lo | x_hi] rows
                # (mixed form: only the last parity_last_blocks blocks; the others keep their plain operands)
                k_par = self.parity_last_blocks
                if k_par is None or i >= len(self.blocks) - k_par:
                    for name, lin in (("qkv", b.attn.qkv), ("proj", b.attn.proj), ("fc1", b.mlp.fc1), ("fc2", b.mlp.fc2)):
                        d[name + "_w3"] = w3(lin.weight, dtype=c)
            elif self.fp8:
                # fp8 tower mode: the four big GEMMs on e4m3 operands (weights per-output-row scaled), LayerNorm as
                # a stand-alone kernel writing fp8 (its output is well scaled; the raw stream is not)
                for name, lin in (("qkv", b.attn.qkv), ("proj", b.attn.proj), ("fc1", b.mlp.fc1), ("fc2", b.mlp.fc2)):
                    d[name + "_w8"], d[name + "_s"] = w8(lin.weight)
            elif self.fuse_layernorm:
                # norm2 folded into fc1 in every block; norm1 folded into qkv from block 1 on (block 0's input comes
                # from the patch-embedding GEMM, which writes no 16-bit copy of the stream)
                d["fc1_f"] = fold_layernorm(b.mlp.fc1.weight, b.mlp.fc1.bias, b.norm2.weight, b.norm2.bias, c)
                if i > 0:
                    d["qkv_f"] = fold_layernorm(b.attn.qkv.weight, b.attn.qkv.bias, b.norm1.weight, b.norm1.bias, c)
            p["blocks"].append(d)
        p["fused"] = self.fuse_layernorm and not self.fp8 and not self.parity
        p["fp8"] = self.fp8
        p["parity"] = self.parity
        if self.parity:
            p["pe_w3"] = w3_patch(pe.weight, c)
        return p

    # ------------------------------------------------------------------ forward
    def embed_patches(self, patches16, B):
        """patches16: f16 [B*P, 3*ps*ps] (from patchify_*).  Returns the f32 residual stream [B*T, D]."""
        p = self.packed()
        D, P = self.embed_dim, self.patch_embed.num_patches
        T = P + 1
        x = torch.empty((B * T, D), dtype=torch.float32, device=patches16.device)
        # (parity mode: patches16 holds [hi | lo | hi] rows, see forward_u8)
        K.gemm(patches16, p["pe_w3"] if p["parity"] else p["pe_w"], p["pe_b"], patch=dict(out=x, pos=p["pos"], tpi=P), split_k=p["parity"])
        K.set_cls_row(x, p["cls"], p["pos"], B, T, D)
        return x

    def run_blocks(self, x, B, want16=True):
        """x: f32 [B*T, D] residual stream (modified in place).  Returns (y32, y16) after the final LN."""
        p = self.packed()
        D, H = self.embed_dim, self.num_heads
        T = self.patch_embed.num_patches + 1
        # V stays row-major (NP = 0): the QKV GEMM stores it like K with 16-B stores and the staged attention kernel
        # transposes it on the way into LDS — cheaper than scattering V^T from the GEMM epilogue (T > 32 rows here)
        NP = 0 if T > 32 else (T + 15) // 16 * 16     # (tiny test geometries fall back to V^T + the direct kernels)
        dev = x.device
        M = B * T
        cdt = p["pe_w"].dtype
        xn = torch.empty((M, D), dtype=cdt, device=dev)
        q = torch.empty((B, H, T, 64), dtype=cdt, device=dev)
        k = torch.empty((B, H, T, 64), dtype=cdt, device=dev)
        vt = torch.empty((B, H, T, 64) if NP == 0 else (B, H, 64, NP), dtype=cdt, device=dev)
        o = torch.empty((M, D), dtype=cdt, device=dev)
        hid = torch.empty((M, p["blocks"][0]["fc1_w"].shape[0]), dtype=cdt, device=dev)
        heads = dict(q=q, k=k, vt=vt, T=T, H=H, part0=0, t_off=0, Tq_cap=T, Tk_cap=T, NP=NP, q_scale=0.125)
        if p.get("parity"):
            return self._run_blocks_parity(p, x, B, T, q, k, vt, heads, NP, want16)
        if p.get("fp8"):
            return self._run_blocks_fp8(p, x, B, T, q, k, vt, heads, NP, want16)
        # Fused LayerNorm (models/vit.py:107-110): the residual GEMMs (proj, fc2) also store the stream in the operand
        # type (``xn`` then holds RAW x, not LN(x)) and the next GEMM applies the LayerNorm to its accumulators.
        fused = p.get("fused", False)
        nblk = len(p["blocks"])
        stats = torch.empty((M, D // 64, 2), dtype=torch.float32, device=dev) if fused else None
        for i, b in enumerate(p["blocks"]):
            if fused and i > 0:
                w_, b_, cs = b["qkv_f"]
                K.gemm(xn, w_, b_, heads=heads, ln=(cs, self.ln_eps, stats))
            else:
                K.layernorm(x, b["n1g"], b["n1b"], self.ln_eps, out16=xn)
                K.gemm(xn, b["qkv_w"], b["qkv_b"], heads=heads)
            K.attention(q, k, vt, o, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=NP)
            if fused:
                K.gemm(o, b["proj_w"], b["proj_b"], out=x, resid=x, out16=xn, ln_stats_out=stats)
                w_, b_, cs = b["fc1_f"]
                K.gemm(xn, w_, b_, out=hid, act=K.ACT_GELU_ERF, ln=(cs, self.ln_eps, stats))
                K.gemm(hid, b["fc2_w"], b["fc2_b"], out=x, resid=x, out16=xn if i + 1 < nblk else None,
                       ln_stats_out=stats if i + 1 < nblk else None)
            else:
                K.gemm(o, b["proj_w"], b["proj_b"], out=x, resid=x)
                K.layernorm(x, b["n2g"], b["n2b"], self.ln_eps, out16=xn)
                K.gemm(xn, b["fc1_w"], b["fc1_b"], out=hid, act=K.ACT_GELU_ERF)
                K.gemm(hid, b["fc2_w"], b["fc2_b"], out=x, resid=x)
        y32 = torch.empty((M, D), dtype=torch.float32, device=dev)
        y16 = xn if want16 else None
        K.layernorm(x, p["norm_g"], p["norm_b"], self.ln_eps, out16=y16, out32=y32)
        return y32, y16

    def _run_blocks_parity(self, p, x, B, T, q, k, vt, heads, NP, want16):
        """Parity precision mode: the same block sequence with every GEMM on error-compensated operands (K tripled):
        LayerNorm and attention write [hi | lo | hi] rows directly (VIDIL_DT_SPLIT3), the GELU output goes through f32
        and vidil_split3_f32.  Returns (y32, y3) with y3 = [M, 3D] split rows of the final LayerNorm (the cross K|V
        projection's operand).  ~3x the MFMA work of the plain path.  Attention (packing.set_parity_attention): "split" / "f32" —
        Q | K | V stay f32 rows of the projection GEMM's output and vidil_attention_f32 reads them in place (split-operand MFMA
        or f32 arithmetic); "16" — the per-head scatter and the 16-bit kernels, Q / K / V and the probabilities rounded to 16 bits."""
        D, H = self.embed_dim, self.num_heads
        dev, cdt = x.device, q.dtype
        M = B * T
        Dh = p["blocks"][0]["fc1_w"].shape[0]
        a3 = torch.empty((M, 3 * D), dtype=cdt, device=dev)
        o3 = torch.empty((M, 3 * D), dtype=cdt, device=dev)
        hid3 = torch.empty((M, 3 * Dh), dtype=cdt, device=dev)
        # (a consumer that takes the K-loop form of the compensated product reads planes hi | lo of its operand rows only, so their
        #  producer need not write the third — decided per CALL from the blocks' actual consumers (the process-wide
        #  switch alone is not enough, e.g. the per-head epilogue with fewer than 8 tokens runs the plain K = 3 Kl product); the
        #  consumers then state a_planes, so a launch that would read an unwritten plane fails instead of computing on it)
        f32_attn, arith = parity_attention_f32(self), parity_attention_arith(self)
        qkv32 = torch.empty((M, 3 * D), dtype=torch.float32, device=dev) if f32_attn else None
        b3 = next((b for b in p["blocks"] if "qkv_w3" in b), None)
        planes = 3
        if b3 is not None and K.split_k_in_loop():
            qkv_kw = dict(out=qkv32) if f32_attn else dict(heads=heads)
            if (K.split_k_serves(a3, b3["qkv_w3"], b3["qkv_b"], **qkv_kw) and K.split_k_serves(o3, b3["proj_w3"], b3["proj_b"], out=x, resid=x)
                    and K.split_k_serves(a3, b3["fc1_w3"], b3["fc1_b"], split3_out=hid3, act=K.ACT_GELU_ERF)
                    and K.split_k_serves(hid3, b3["fc2_w3"], b3["fc2_b"], out=x, resid=x)):
                planes = 2
        import os
        if planes == 2 and os.environ.get("VIDIL_POISON_SPLIT3") == "1":       # (developer: NaNs in the unwritten third planes —
            for _b in (a3, o3, hid3,):                               #  any consumer that reads one shows up at once)
                _b[:, 2 * (_b.shape[1] // 3):] = float("nan")
        plain = [b for b in p["blocks"] if "qkv_w3" not in b]
        if plain:       # mixed form: the leading blocks on plain 16-bit operands (unfused: LayerNorm kernel + plain GEMM)
            xn = torch.empty((M, D), dtype=cdt, device=dev)
            o = torch.empty((M, D), dtype=cdt, device=dev)
            hid = torch.empty((M, Dh), dtype=cdt, device=dev)
        for b in p["blocks"]:
            if "qkv_w3" not in b:
                K.layernorm(x, b["n1g"], b["n1b"], self.ln_eps, out16=xn)
                K.gemm(xn, b["qkv_w"], b["qkv_b"], heads=heads)
                K.attention(q, k, vt, o, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=NP)
                K.gemm(o, b["proj_w"], b["proj_b"], out=x, resid=x)
                K.layernorm(x, b["n2g"], b["n2b"], self.ln_eps, out16=xn)
                K.gemm(xn, b["fc1_w"], b["fc1_b"], out=hid, act=K.ACT_GELU_ERF)
                K.gemm(hid, b["fc2_w"], b["fc2_b"], out=x, resid=x)
                continue
            K.layernorm(x, b["n1g"], b["n1b"], self.ln_eps, out16=a3, split3=True, planes=planes)
            if f32_attn:    # Q | K | V stay f32 and row-major; the f32 attention reads them in place (no per-head scatter)
                K.gemm(a3, b["qkv_w3"], b["qkv_b"], out=qkv32, split_k=True, a_planes=planes)
                K.attention_f32(qkv32[:, :D], qkv32[:, D:2 * D], qkv32[:, 2 * D:], o3, Bq=B, H=H, Nq=T, Nk=T, arith=arith, planes=planes)
            else:
                K.gemm(a3, b["qkv_w3"], b["qkv_b"], heads=heads, split_k=True, a_planes=planes)
                K.attention(q, k, vt, o3, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=NP, split3=True)      # (writes all three planes)
            K.gemm(o3, b["proj_w3"], b["proj_b"], out=x, resid=x, split_k=True, a_planes=planes if f32_attn else 3)
            K.layernorm(x, b["n2g"], b["n2b"], self.ln_eps, out16=a3, split3=True, planes=planes)
            # (fc1 + erf-GELU in f32, handed to fc2 as [hi | lo | hi] rows by the GEMM's own epilogue: no f32 round trip)
            K.gemm(a3, b["fc1_w3"], b["fc1_b"], split3_out=hid3, act=K.ACT_GELU_ERF, split_k=True, split3_planes=planes, a_planes=planes)
            K.gemm(hid3, b["fc2_w3"], b["fc2_b"], out=x, resid=x, split_k=True, a_planes=planes)
        y32 = torch.empty((M, D), dtype=torch.float32, device=dev)
        # (the image tokens leave this module: their consumer — BertModel.project_cross_kv, any epilogue, any token count — is not
        #  known here, so all three planes are written: one launch per forward)
        K.layernorm(x, p["norm_g"], p["norm_b"], self.ln_eps, out16=a3 if want16 else None, out32=y32, split3=True, planes=3)
        return y32, (a3 if want16 else None)

    def _run_blocks_fp8(self, p, x, B, T, q, k, vt, heads, NP, want16):
        """fp8 tower mode (BASELINE config 5): LN -> fp8, QKV / proj / fc1 / fc2 on e4m3 operands at twice the 16-bit
        MFMA rate, attention on the 16-bit companion type writing fp8, f32 residual stream.  NOT a parity mode: e4m3
        carries 3 mantissa bits (tests/test_fp8_gpu.py states the measured deviation from the fp32 oracle)."""
        D, H = self.embed_dim, self.num_heads
        dev = x.device
        M = B * T
        xn8 = torch.empty((M, D), dtype=FP8, device=dev)
        o8 = torch.empty((M, D), dtype=FP8, device=dev)
        hid8 = torch.empty((M, p["blocks"][0]["fc1_w8"].shape[0]), dtype=FP8, device=dev)
        for b in p["blocks"]:
            K.layernorm(x, b["n1g"], b["n1b"], self.ln_eps, out16=xn8)
            K.gemm(xn8, b["qkv_w8"], b["qkv_b"], heads=heads, w_scale=b["qkv_s"])
            K.attention(q, k, vt, o8, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=NP)
            K.gemm(o8, b["proj_w8"], b["proj_b"], out=x, resid=x, w_scale=b["proj_s"], dtype16=q.dtype)
            K.layernorm(x, b["n2g"], b["n2b"], self.ln_eps, out16=xn8)
            K.gemm(xn8, b["fc1_w8"], b["fc1_b"], out=hid8, act=K.ACT_GELU_ERF, w_scale=b["fc1_s"], dtype16=q.dtype)
            K.gemm(hid8, b["fc2_w8"], b["fc2_b"], out=x, resid=x, w_scale=b["fc2_s"], dtype16=q.dtype)
        y32 = torch.empty((M, D), dtype=torch.float32, device=dev)
        y16 = torch.empty((M, D), dtype=q.dtype, device=dev) if want16 else None
        K.layernorm(x, p["norm_g"], p["norm_b"], self.ln_eps, out16=y16, out32=y32)
        return y32, y16

    def forward_both(self, x):
        """x f32 [B,3,S,S] on the GPU -> (f32 [B,T,D], f16 [B*T,D])."""
        require_cuda(x, "VisionTransformer.forward")
        B = x.shape[0]
        ps = self.patch_embed.patch_size[0]
        patches = K.patchify_f32(x.contiguous().float(), ps, dtype=self.cdt, split3=self.parity)
        xr = self.embed_patches(patches, B)
        y32, y16 = self.run_blocks(xr, B)
        return y32.view(B, -1, self.embed_dim), y16

    def forward_u8(self, frames_u8, mean, std):
        """uint8 [B,S,S,3] frames (already S x S) with fused /255 + normalise."""
        require_cuda(frames_u8, "VisionTransformer.forward_u8")
        B = frames_u8.shape[0]
        ps = self.patch_embed.patch_size[0]
        patches = K.patchify_u8(frames_u8.contiguous(), ps, mean, std, dtype=self.cdt, split3=self.parity)
        xr = self.embed_patches(patches, B)
        y32, y16 = self.run_blocks(xr, B)
        return y32.view(B, -1, self.embed_dim), y16

    def forward(self, x, register_blk=-1):
        return self.forward_both(x)[0]


def interpolate_pos_embed(pos_embed_checkpoint, visual_encoder):
    """Resize a checkpoint's position grid to this encoder's (reference: models/vit.py:281-305)."""
    width = pos_embed_checkpoint.shape[-1]
    num_patches = visual_encoder.patch_embed.num_patches
    extra = visual_encoder.pos_embed.shape[-2] - num_patches
    old = int((pos_embed_checkpoint.shape[-2] - extra) ** 0.5)
    new = int(num_patches ** 0.5)
    if old == new:
        return pos_embed_checkpoint
    keep = pos_embed_checkpoint[:, :extra]
    grid = pos_embed_checkpoint[:, extra:].reshape(-1, old, old, width).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(new, new), mode="bicubic", align_corners=False)
    grid = grid.permute(0, 2, 3, 1).flatten(1, 2)
    print("reshape position embedding from %d to %d" % (old ** 2, new ** 2))
    return torch.cat((keep, grid), dim=1)
