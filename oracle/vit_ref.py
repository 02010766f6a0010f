"""oracle/vit_ref.py — TEST INFRASTRUCTURE.  fp32 restatement of the BLIP ViT
(reference: models/vit.py; timm PatchEmbed as used at models/vit.py:144-145).

All functions take a flat ``state_dict`` with the reference's checkpoint key
names (``<prefix>blocks.3.attn.qkv.weight`` …) so the same dict drives the
reference module, this oracle and the HIP path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _ln(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def patch_embed(sd, prefix, img, patch):
    """models/vit.py:182 -> timm PatchEmbed: conv(k=s=patch), flatten(2), transpose(1,2)."""
    y = F.conv2d(img, sd[prefix + "patch_embed.proj.weight"], sd.get(prefix + "patch_embed.proj.bias"), stride=patch)
    return y.flatten(2).transpose(1, 2)


def attention(sd, p, x, heads):
    """models/vit.py:70-86."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[p + "qkv.weight"], sd.get(p + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q @ k.transpose(-2, -1)) * ((C // heads) ** -0.5)
    att = att.softmax(dim=-1)
    y = (att @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"])


def mlp(sd, p, x):
    """models/vit.py:35-41 (exact erf GELU)."""
    h = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    return F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def block(sd, p, x, heads, eps=1e-6):
    """models/vit.py:107-110 (pre-LN residual block, LN eps 1e-6 from :142)."""
    x = x + attention(sd, p + "attn.", _ln(x, sd, p + "norm1", eps), heads)
    x = x + mlp(sd, p + "mlp.", _ln(x, sd, p + "norm2", eps))
    return x


def vit_forward(sd, img, *, prefix="visual_encoder.", patch=16, depth=12, heads=12, eps=1e-6, return_blocks=False):
    """models/vit.py:180-194.  img [B,3,S,S] f32 -> [B, 1+(S/patch)^2, width]."""
    B = img.shape[0]
    x = patch_embed(sd, prefix, img, patch)
    cls = sd[prefix + "cls_token"].expand(B, -1, -1)
    x = torch.cat((cls, x), dim=1)
    x = x + sd[prefix + "pos_embed"][:, : x.size(1), :]
    per_block = [x]
    for i in range(depth):
        x = block(sd, f"{prefix}blocks.{i}.", x, heads, eps)
        if return_blocks:
            per_block.append(x)
    x = _ln(x, sd, prefix + "norm", eps)
    return (x, per_block) if return_blocks else x


def interpolate_pos_embed(pos_embed_ckpt, num_patches, num_extra=1):
    """models/vit.py:281-305: bicubic resize of the patch position grid, extra tokens kept."""
    width = pos_embed_ckpt.shape[-1]
    orig = int((pos_embed_ckpt.shape[-2] - num_extra) ** 0.5)
    new = int(num_patches ** 0.5)
    if orig == new:
        return pos_embed_ckpt
    extra = pos_embed_ckpt[:, :num_extra]
    grid = pos_embed_ckpt[:, num_extra:].reshape(-1, orig, orig, width).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(new, new), mode="bicubic", align_corners=False)
    grid = grid.permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((extra, grid), dim=1)
