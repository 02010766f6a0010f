"""Generate the golden vectors under tests/golden/ from the REFERENCE's own modules.

Run in the build container only (needs /root/reference and `transformers`):

    python tests/golden/make_golden.py

It imports the reference's models/vit.py and models/med.py through oracle/ref_shim.py and
HF's CLIPModel, builds SMALL seeded configurations of them (so a whole state_dict fits in a
few MB; widths keep head_dim 64, the only head size on the path), rounds every weight to an
f16-representable value (so the f16 MFMA path and the fp32 reference share bit-identical
weights), runs the reference and stores inputs, weights and outputs as .npz.  The files are
data: tensors in, tensors out.  Nothing here is needed at test time.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shim, tokens_ref  # noqa: E402


def f16_exact_(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)   # non-trivial LN gains / biases
            p.copy_(p.to(torch.float16).to(torch.float32))


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().numpy().astype(np.float16) for k, v in module.state_dict().items()
            if v.dtype == torch.float32}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB")


def main():
    vit_mod, med_mod = ref_shim.load()
    torch.manual_seed(1234)

    # ---------------------------------------------------------------- ViT (models/vit.py)
    vit = vit_mod.VisionTransformer(img_size=64, patch_size=16, embed_dim=256, depth=2, num_heads=4).eval()
    f16_exact_(vit, 1)
    x = torch.randn(3, 3, 64, 64)
    blocks = []
    hooks = [b.register_forward_hook(lambda m, i, o: blocks.append(o.detach().numpy())) for b in vit.blocks]
    with torch.no_grad():
        y = vit(x)
    for h in hooks:
        h.remove()
    arrays = {"w/" + k: v for k, v in sd_np(vit, "visual_encoder.").items()}
    save("vit_small.npz", x=x.numpy(), y=y.numpy(), block0=blocks[0], block1=blocks[1], **arrays)

    # ---------------------------------------------------------------- MED decoder (models/med.py)
    cfg = ref_shim.med_config(encoder_width=256)
    cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size = 256, 4, 512
    cfg.num_hidden_layers, cfg.vocab_size, cfg.max_position_embeddings = 2, 512, 64
    dec = med_mod.BertLMHeadModel(cfg).eval()
    f16_exact_(dec, 2)
    enc = y.detach()                                  # [3, 17, 256] image tokens
    enc3 = enc.repeat_interleave(2, dim=0)            # 2 beams per image -> 6 rows
    ones = torch.ones(enc3.shape[:2], dtype=torch.long)
    ids = torch.tensor([[510, 7, 8, 9]] * 6)
    with torch.no_grad():
        o0 = dec(ids, attention_mask=torch.ones_like(ids), encoder_hidden_states=enc3, encoder_attention_mask=ones,
                 return_dict=True, is_decoder=True, use_cache=True)
        beam_idx = torch.tensor([1, 0, 2, 2, 5, 4])
        past = dec._reorder_cache(o0.past_key_values, beam_idx)
        nxt = torch.tensor([[11], [12], [13], [14], [15], [16]])
        ids1 = torch.cat([ids[beam_idx], nxt], dim=1)
        o1 = dec(ids1[:, -1:], attention_mask=torch.ones_like(ids1), encoder_hidden_states=enc3,
                 encoder_attention_mask=ones, past_key_values=past, return_dict=True, is_decoder=True, use_cache=True)
        nxt2 = torch.tensor([[21], [22], [23], [24], [25], [26]])
        ids2 = torch.cat([ids1, nxt2], dim=1)
        o2 = dec(ids2[:, -1:], attention_mask=torch.ones_like(ids2), encoder_hidden_states=enc3,
                 encoder_attention_mask=ones, past_key_values=o1.past_key_values, return_dict=True, is_decoder=True,
                 use_cache=True)
    arrays = {"w/" + k: v for k, v in sd_np(dec, "text_decoder.").items()}
    save("med_decoder_small.npz", enc=enc.numpy(), ids=ids.numpy(), beam_idx=beam_idx.numpy(), ids1=ids1.numpy(),
         ids2=ids2.numpy(), logits0=o0.logits[:, -1].numpy(), logits1=o1.logits[:, -1].numpy(),
         logits2=o2.logits[:, -1].numpy(), k_cache_l1=o0.past_key_values[1][0].numpy(), **arrays)

    # ---------------------------------------------------------------- MED ITM encoder (models/blip_itm.py:51-57)
    encm = med_mod.BertModel(cfg, add_pooling_layer=False).eval()
    f16_exact_(encm, 3)
    itm_head = torch.nn.Linear(256, 2)
    f16_exact_(itm_head, 4)
    tid = torch.randint(20, 500, (3, 35))
    tid[:, 0] = 101 % 512
    lens = [35, 9, 20]
    am = torch.zeros(3, 35, dtype=torch.long)
    for i, L in enumerate(lens):
        am[i, :L] = 1
        tid[i, L:] = 0
    with torch.no_grad():
        o = encm(tid, attention_mask=am, encoder_hidden_states=enc, encoder_attention_mask=torch.ones(3, 17, dtype=torch.long),
                 return_dict=True)
        itm = itm_head(o.last_hidden_state[:, 0, :])
    arrays = {"w/" + k: v for k, v in sd_np(encm, "text_encoder.").items()}
    arrays["w/itm_head.weight"] = itm_head.weight.detach().numpy().astype(np.float16)
    arrays["w/itm_head.bias"] = itm_head.bias.detach().numpy().astype(np.float16)
    save("med_itm_small.npz", enc=enc.numpy(), ids=tid.numpy(), mask=am.numpy(), itm=itm.numpy(),
         hidden=o.last_hidden_state.numpy(), **arrays)

    # ---------------------------------------------------------------- CLIP (transformers CLIPModel)
    import transformers
    from transformers import CLIPConfig, CLIPModel

    ccfg = CLIPConfig(
        vision_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                           image_size=64, patch_size=32),
        text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                         vocab_size=1000, max_position_embeddings=16, eos_token_id=999, bos_token_id=998, pad_token_id=1),
        projection_dim=128)
    clip = CLIPModel(ccfg).eval()
    f16_exact_(clip, 5)
    pv = torch.randn(4, 3, 64, 64)
    tids = torch.randint(2, 990, (5, 10))
    tids[:, 0] = 998
    tids[:, -1] = 999
    tids[2, 6:] = 999
    tam = torch.ones(5, 10, dtype=torch.long)
    tam[2, 7:] = 0
    with torch.no_grad():
        out = clip(input_ids=tids, attention_mask=tam, pixel_values=pv)
    arrays = {"w/" + k: v for k, v in sd_np(clip).items()}
    save("clip_small.npz", pixel_values=pv.numpy(), input_ids=tids.numpy(), attention_mask=tam.numpy(),
         image_embeds=out.image_embeds.numpy(), text_embeds=out.text_embeds.numpy(),
         transformers_version=np.array(transformers.__version__), **arrays)

    # ---------------------------------------------------------------- ontology filter (run_visual_tokenization.py:369-396)
    root = os.path.join(ref_shim.REFERENCE_ROOT, "visual_token_ontology")
    sizes = {}
    for name in ("vg", "vg_tencent"):
        ont = tokens_ref.load_ontology(root, name)
        sizes[name] = {k: dict(n=len(v), distinct=len(set(v)), first=v[:3], last=v[-3:]) for k, v in ont.items()}
    with open(os.path.join(HERE, "ontology_sizes.json"), "w") as f:
        json.dump(sizes, f, indent=1)
    print(json.dumps({k: {c: v["n"] for c, v in s.items()} for k, s in sizes.items()}))


if __name__ == "__main__":
    main()
