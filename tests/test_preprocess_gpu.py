"""GPU parity of the frame resize (vidil_resample_u8 through vidil_amd/preprocess.py): bit-exact against the
oracle's restatement of Pillow and against the golden bytes Pillow itself produced."""
import os

import numpy as np
import pytest
import torch

from common import ROOT
from oracle import resize_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_golden_pillow_bytes_blip_and_clip():
    from vidil_amd.preprocess import blip_frames, clip_frames

    g = np.load(os.path.join(ROOT, "tests", "golden", "resize_golden.npz"))
    for key in sorted(k[:-3] for k in g.files if k.endswith("_in")):
        img, S = g[key + "_in"], int(g[key + "_S"])
        x = torch.from_numpy(np.stack([img, img[::-1].copy()])).to(DEV)        # batch of 2 (second flipped)
        b = blip_frames(x, S).cpu().numpy()
        c = clip_frames(x, S).cpu().numpy()
        assert np.array_equal(b[0], g[key + "_blip"]), key
        assert np.array_equal(c[0], g[key + "_clip"]), key
        assert np.array_equal(b[1], R.blip_process_frame_u8(img[::-1].copy(), S)), key
        assert np.array_equal(c[1], R.clip_process_frame_u8(img[::-1].copy(), S)), key


@pytest.mark.parametrize("H,W", [(360, 640), (240, 320), (100, 130), (480, 270), (224, 300), (300, 224), (37, 53), (224, 224)])
def test_full_size_frames_vs_oracle(H, W):
    from vidil_amd.preprocess import blip_frames, clip_frames

    rng = np.random.default_rng(H * 1000 + W)
    frames = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    x = torch.from_numpy(frames).to(DEV)
    b = blip_frames(x, 224).cpu().numpy()
    c = clip_frames(x, 224).cpu().numpy()
    assert b.shape == (3, 224, 224, 3) and c.shape == (3, 224, 224, 3)
    for i in range(3):
        assert np.array_equal(b[i], R.blip_process_frame_u8(frames[i], 224))
        assert np.array_equal(c[i], R.clip_process_frame_u8(frames[i], 224))


def test_blip_384_and_error_paths():
    from vidil_amd import kernels as K
    from vidil_amd.preprocess import blip_frames

    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (2, 270, 480, 3), dtype=np.uint8)
    out = blip_frames(torch.from_numpy(frames).to(DEV), 384).cpu().numpy()
    for i in range(2):
        assert np.array_equal(out[i], R.blip_process_frame_u8(frames[i], 384))
    with pytest.raises(K.VidilHipError):
        blip_frames(torch.zeros(2, 10, 10, 3, device=DEV), 224)           # not uint8
    with pytest.raises(Exception):
        blip_frames(torch.zeros(2, 10, 10, 3, dtype=torch.uint8), 224)    # not on the GPU: no CPU fallback


def test_engines_accept_non_square_frames_end_to_end():
    """CapFilt + visual tokens on 90x160 frames == the same engines fed the oracle-resized frames."""
    from vidil_amd.clip import CLIPModel
    from vidil_amd.preprocess import blip_frames, clip_frames
    from vidil_amd.vit import VisionTransformer

    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (4, 90, 160, 3), dtype=np.uint8)
    x = torch.from_numpy(frames).to(DEV)
    ref_b = torch.from_numpy(np.stack([R.blip_process_frame_u8(f, 224) for f in frames])).to(DEV)
    ref_c = torch.from_numpy(np.stack([R.clip_process_frame_u8(f, 224) for f in frames])).to(DEV)
    torch.manual_seed(0)
    vit = VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=2, num_heads=12).to(DEV)
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    y1, _ = vit.forward_u8(blip_frames(x, 224), mean, std)
    y2, _ = vit.forward_u8(ref_b, mean, std)
    assert torch.equal(y1, y2)
    clip = CLIPModel().eval().to(DEV)
    assert torch.equal(clip.encode_image_u8(clip_frames(x, 224)), clip.encode_image_u8(ref_c))


def test_clip_processor_and_model_call_signatures_of_the_reference():
    """run_visual_tokenization.py:138-142: processor(text=..., images=[PIL ...], return_tensors='pt', padding=True)
    .to(device) -> model(**inputs).image_embeds; images of mixed sizes, resized on the GPU like CLIPImageProcessor."""
    from PIL import Image

    from vidil_amd.clip import CLIPConfig, CLIPModel, CLIPProcessor, CLIPTextConfig, CLIPVisionConfig
    from vidil_amd.tokenizer import Encoding

    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in ((90, 160, 3), (120, 100, 3), (90, 160, 3), (64, 64, 3))]
    imgs[2] = imgs[0].copy()

    def fake_bpe(texts, return_tensors="pt", padding=True, truncation=True, **_):
        ids = torch.full((len(texts), 6), 299, dtype=torch.long)
        ids[:, 0] = 298
        ids[:, 1] = torch.arange(len(texts)) + 5
        return Encoding(input_ids=ids, attention_mask=torch.ones_like(ids))

    proc = CLIPProcessor(tokenizer=fake_bpe, image_size=64)
    inputs = proc(text=["hello world"], images=[Image.fromarray(a) for a in imgs], return_tensors="pt", padding=True).to(DEV)
    pv = inputs["pixel_values"]
    assert pv.dtype == torch.uint8 and tuple(pv.shape) == (4, 64, 64, 3) and pv.is_cuda
    for i, a in enumerate(imgs):           # byte-identical with the Pillow path of HF's CLIPImageProcessor
        assert np.array_equal(pv[i].cpu().numpy(), R.clip_process_frame_u8(a, 64)), i
    torch.manual_seed(0)
    cfg = CLIPConfig(CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=4,
                                      image_size=64, patch_size=16),
                     CLIPTextConfig(vocab_size=300, hidden_size=256, intermediate_size=512, num_hidden_layers=1,
                                    num_attention_heads=4, max_position_embeddings=16, eos_token_id=299), 64)
    model = CLIPModel(cfg).eval().to(DEV)
    out = model(**inputs)
    assert tuple(out.image_embeds.shape) == (4, 64) and tuple(out.text_embeds.shape) == (1, 64)
    assert torch.allclose(out.image_embeds.norm(dim=-1), torch.ones(4, device=DEV), atol=1e-5)
    assert torch.equal(out.image_embeds[0], out.image_embeds[2])        # identical frames -> identical embeddings
    # the HF-style f32 NCHW pixel_values entry point gives the same embeddings up to the f16 rounding of the input
    x = (pv.float().permute(0, 3, 1, 2) / 255.0 - torch.tensor((0.48145466, 0.4578275, 0.40821073), device=DEV).view(1, 3, 1, 1)) \
        / torch.tensor((0.26862954, 0.26130258, 0.27577711), device=DEV).view(1, 3, 1, 1)
    out2 = model(pixel_values=x.contiguous())
    assert (out2.image_embeds - out.image_embeds).abs().max().item() < 2e-3
