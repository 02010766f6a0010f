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
