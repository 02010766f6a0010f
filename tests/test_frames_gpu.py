"""'clip-kmeans' frame selection (data/video_pretrain_dataset.py:190-216) on the HIP CLIP tower vs the CPU oracle."""
import numpy as np
import pytest
import torch

from common import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _small_clip():
    from vidil_amd.clip import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig

    sd, _ = load_golden("clip_small.npz")
    cfg = CLIPConfig(CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                      image_size=64, patch_size=32),
                     CLIPTextConfig(vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                    num_attention_heads=4, max_position_embeddings=16, eos_token_id=999), 128)
    m = CLIPModel(cfg)
    own = m.state_dict()
    m.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return m.to(DEV).eval(), sd


def _scenes_video(n_scenes=4, per_scene=6, S=64, seed=3):
    """A 'video' of n_scenes visually distinct shots (a random base image each) with small per-frame noise."""
    rng = np.random.default_rng(seed)
    frames, scene = [], []
    for s in range(n_scenes):
        base = rng.integers(0, 256, size=(S, S, 3)).astype(np.int16)
        for _ in range(per_scene):
            frames.append(np.clip(base + rng.integers(-3, 4, size=base.shape), 0, 255).astype(np.uint8))
            scene.append(s)
    return np.stack(frames), np.array(scene)


def test_pooled_output_vs_oracle_and_kmeans_selection():
    from oracle import clip_ref
    from vidil_amd import frames as fr

    m, sd = _small_clip()
    video, scene = _scenes_video()
    dev_video = torch.from_numpy(video).to(DEV)
    cand = np.arange(len(video), step=2, dtype=int)

    got = fr.clip_pooled(m, dev_video[torch.from_numpy(cand).to(DEV)])
    ref = clip_ref.pooled_output({k: v.float() for k, v in sd.items()}, clip_ref.preprocess_u8(video[cand]), layers=2, heads=4,
                                 patch=32).numpy()
    assert got.shape == ref.shape == (len(cand), 256)
    assert np.abs(got - ref).max() < 5e-3 * max(1.0, np.abs(ref).max())        # f16 operands vs the fp32 oracle

    # the selection: same k-means call, same random stream -> same frames as the oracle embeddings give
    picked = fr.clip_kmeans_indices(m, dev_video, 4, downsample_ratio=2, np_random=np.random.RandomState(11))
    want = fr.kmeans_pick(ref, 4, cand, np_random=np.random.RandomState(11))
    assert picked == want
    assert sorted(scene[picked].tolist()) == [0, 1, 2, 3]                         # one frame per shot
    # through frame_indices, as the loader calls it
    sel = fr.frame_indices(len(video), 4, "clip-kmeans",
                           clip_select=lambda n: fr.clip_kmeans_indices(m, dev_video, n, np_random=np.random.RandomState(11)))
    assert sel == picked


def test_pooled_output_resizes_like_the_clip_processor():
    """Frames that are not S x S take the CLIPProcessor path (shortest edge -> S, centre crop) on the GPU."""
    from vidil_amd import frames as fr
    from vidil_amd.preprocess import clip_frames

    m, _ = _small_clip()
    rng = np.random.default_rng(0)
    video = torch.from_numpy(rng.integers(0, 256, size=(5, 96, 128, 3), dtype=np.uint8)).to(DEV)
    a = fr.clip_pooled(m, video)
    b = m.pooled_image_u8(clip_frames(video, 64)).float().cpu().numpy()
    assert a.shape == (5, 256) and np.array_equal(a, b)
