"""Which frames of a video feed the hot path — the index arithmetic of the reference's decord loader
(`data/video_pretrain_dataset.py:116-188`, same code again at `:305-377` and `:477-534`), with the
decoder itself left outside (decord is not part of this path; callers hand over decoded uint8 frames).

``frame_indices`` reproduces every `frm_sampling_strategy` of the reference on the same random streams
(Python's ``random`` for 'rand' / 'headtail', ``numpy.random`` for 'nlvl_rand' and the k-means pick), so a
run seeded like the reference (`run_video_CapFilt.py:230`: seed + rank) samples the same frames.
``clip_kmeans_indices`` is the 'clip-kmeans' strategy (`:190-216`): CLIP ``pooler_output`` of every
``downsample_ratio``-th frame on the HIP vision tower, scikit-learn ``KMeans(num_frm, random_state=0)`` on
the host exactly as the reference calls it, one random member per cluster.
"""
from __future__ import annotations

import random as _py_random

import numpy as np
import torch

STRATEGIES = ("uniform", "nlvl_uniform", "nlvl_rand", "rand", "headtail", "clip-kmeans")


def clip_range(vlen, start_time=None, end_time=None, fps=-1):
    """(start_idx, end_idx) of the sampled span (`data/video_pretrain_dataset.py:134-141`)."""
    if start_time or end_time:
        assert fps > 0, "must provide video fps if specifying start and end time."
        return min(int(start_time * fps), vlen), min(int(end_time * fps), vlen)
    return 0, vlen


def frame_indices(vlen, num_frm, strategy="uniform", *, start_time=None, end_time=None, fps=-1,
                  py_random=None, np_random=None, clip_select=None):
    """Frame numbers (ascending, python ints) the reference would decode from a ``vlen``-frame video.

    py_random / np_random: objects with ``sample`` / ``randint`` (default: the global ``random`` module and
    ``numpy.random``, the streams the reference draws from).  clip_select: callable(num_frm) -> indices for
    'clip-kmeans' (see clip_kmeans_indices)."""
    py_random = _py_random if py_random is None else py_random
    np_random = np.random if np_random is None else np_random
    start_idx, end_idx = clip_range(vlen, start_time, end_time, fps)
    step = vlen / num_frm                                   # NOT (end-start)/num_frm: the reference divides vlen
    if strategy == "uniform":
        # numpy evaluates an integer arange with a float step as start + i * (int(start+step) - int(start))
        idx = np.arange(start_idx, end_idx, step, dtype=int)
    elif strategy == "nlvl_uniform":
        idx = np.arange(start_idx, end_idx, step).astype(int)
    elif strategy == "nlvl_rand":
        idx = np.arange(start_idx, end_idx, step).astype(int)
        strides = [idx[i] - idx[i - 1] for i in range(1, len(idx))] + [vlen - idx[-1]]
        idx = idx + np.array([np_random.randint(0, s) for s in strides])
    elif strategy == "rand":
        idx = sorted(py_random.sample(range(vlen), num_frm))
    elif strategy == "headtail":
        head = sorted(py_random.sample(range(vlen // 2), num_frm // 2))
        tail = sorted(py_random.sample(range(vlen // 2, vlen), num_frm // 2))
        idx = head + tail
    elif strategy == "clip-kmeans":
        if clip_select is None:
            raise ValueError("frame_indices: 'clip-kmeans' needs clip_select (see clip_kmeans_indices)")
        idx = clip_select(num_frm)
    else:
        raise NotImplementedError("Invalid sampling strategy {} ".format(strategy))
    return [int(i) for i in idx]


@torch.no_grad()
def clip_pooled(clip_model, frames_u8, batch=512):
    """uint8 [n,H,W,3] frames (device) -> f32 [n,D] numpy: HF ``CLIPModel(...).pooler_output`` of the vision
    tower (post_layernorm of the class token, before the projection), CLIPProcessor resize/crop on the GPU."""
    from .preprocess import clip_frames

    S = clip_model.config.vision_config.image_size
    out = []
    for i in range(0, frames_u8.shape[0], batch):
        out.append(clip_model.pooled_image_u8(clip_frames(frames_u8[i:i + batch], S)).float().cpu())
    return torch.cat(out).numpy()


def kmeans_pick(embeddings, num_frm, candidates, np_random=None):
    """`data/video_pretrain_dataset.py:203-214`: KMeans(num_frm, random_state=0) on the embeddings, one random
    member of every cluster (cluster order 0..num_frm-1 = order of the random draws), sorted."""
    from sklearn.cluster import KMeans

    np_random = np.random if np_random is None else np_random
    labels = KMeans(n_clusters=num_frm, random_state=0).fit(embeddings).labels_
    picked = []
    for c in range(num_frm):
        members = np.where(labels == c)[0]
        picked.append(int(candidates[np_random.choice(members)]))
    return sorted(picked)


def clip_kmeans_indices(clip_model, video_u8, num_frm, downsample_ratio=2, np_random=None):
    """'clip-kmeans' on a fully decoded video: uint8 [vlen,H,W,3] on the device -> sorted frame numbers."""
    vlen = video_u8.shape[0]
    candidates = np.arange(vlen, step=downsample_ratio, dtype=int)
    emb = clip_pooled(clip_model, video_u8[torch.from_numpy(candidates).to(video_u8.device)])
    return kmeans_pick(emb, num_frm, candidates, np_random)
