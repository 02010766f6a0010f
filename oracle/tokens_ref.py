"""oracle/tokens_ref.py — TEST INFRASTRUCTURE.  Restatement of the host-side
string/index logic of the reference's visual tokenization
(run_visual_tokenization.py) and of CapFilt's caption post-processing
(run_video_CapFilt.py), plain Python / numpy.
"""
from __future__ import annotations

import json
import os
from collections import defaultdict

import numpy as np

# run_visual_tokenization.py:471-472
OMIT_KEYWORDS = ['media player', 'video', 'playing video', 'audio', 'sound', 'taking video', 'water mark',
                 'water marked', 'watermark', 'watermarks', 'for sale in', 'sold from', 'stock', 'sold on',
                 'by viewers', 'are provided by', 'are posted on', 'for more', 'tag with', 'stream from',
                 'viewed from', 'showing video of', 'are on at', 'shuttlecock', 'shutter', 'shutter is white',
                 'shutters have bones', 'tape is looped', 'bliss wants you', 'thumbnail', 'technique']

CATEGORIES = ("objects", "attributes", "scenes", "verbs")

ONTOLOGY_FILES = {  # run_visual_tokenization.py:369-381
    "vg": dict(objects="vg/openimage_classes_all_cleaned_fictional_characters.json",
               attributes="vg/vg_original_attributes_synsets_keys_cleaned_remove_similar0.9.json",
               scenes="vg/place365_ontology.json",
               verbs="vg/vg_srl_selected_object_synsets_keys_remove_similar0.9.json"),
    "vg_tencent": dict(objects="vg_tencent/tencent_ml_images_objects.json",
                       attributes="vg_tencent/vg_original_attributes_synsets_keys_cleaned_remove_similar0.9.json",
                       scenes="vg/place365_ontology.json",
                       verbs="vg_tencent/vg_srl_selected_object_synsets_keys_remove_similar0.9.json"),
}


def filter_ontology(objects, attributes, scenes, verbs):
    """run_visual_tokenization.py:383-396, including its quirk: ``attribute_texts`` is
    mutated while being iterated, so the element following each removed one is skipped."""
    objects, attributes, scenes = list(objects), list(attributes), list(scenes)
    verbs = list(verbs.keys()) if isinstance(verbs, dict) else list(verbs)
    i = 0
    while i < len(attributes):          # == `for key in attributes: if key in objects: attributes.remove(key)`
        key = attributes[i]
        if key in objects:
            attributes.remove(key)      # removes the FIRST occurrence, like list.remove
        i += 1
    for key in OMIT_KEYWORDS:
        for lst in (objects, attributes, scenes, verbs):
            if key in lst:
                lst.remove(key)
    return dict(objects=objects, attributes=attributes, scenes=scenes, verbs=verbs)


def load_ontology(root, name="vg"):
    files = ONTOLOGY_FILES[name]
    raw = {k: json.load(open(os.path.join(root, v))) for k, v in files.items()}
    return filter_ontology(raw["objects"], raw["attributes"], raw["scenes"], raw["verbs"])


def prompt_texts(texts, version="v1"):
    """run_visual_tokenization.py:56-80."""
    if version == "v0":
        return list(texts)
    return [f"A photo of {t}" for t in texts]


def frame_topk_indices(scores_row, topk):
    """run_visual_tokenization.py:306: np.argsort(frm_score)[::-1][:topk]."""
    return np.argsort(scores_row)[::-1][:topk]


def aggregate_frame_tokens(frame_tokens):
    """run_visual_tokenization.py:173-187: count texts rank-major (rank outer, frame inner),
    stable sort by count descending, keep the first top-k."""
    keys = list(frame_tokens[0].keys())
    out = {k: [] for k in keys}
    topk = len(frame_tokens[0]["objects"])
    for key in keys:
        if frame_tokens[0][key] == []:
            continue
        count = defaultdict(int)
        for j in range(topk):
            for fr in frame_tokens:
                count[fr[key][j]] += 1
        cand = sorted(count.items(), key=lambda x: x[1], reverse=True)
        out[key] = [t for t, _ in cand[:topk]]
    return out


def visual_tokens_from_scores(video_ids, captions, scores_by_cat, texts_by_cat, num_frm, topk):
    """run_visual_tokenization.py:267-312 given the per-category score matrices
    [N_videos*num_frm, N_cat] (numpy f32)."""
    out = {vid: {"frame_tokens": [dict() for _ in range(num_frm)], "caption": captions[i]}
           for i, vid in enumerate(video_ids)}
    for key in CATEGORIES:
        sm = scores_by_cat[key].reshape(len(video_ids), num_frm, -1)
        for j, vid in enumerate(video_ids):
            for f in range(num_frm):
                inds = frame_topk_indices(sm[j][f], topk)
                out[vid]["frame_tokens"][f][key] = [texts_by_cat[key][ii] for ii in inds]
    for vid, obj in out.items():
        obj["aggregated_tokens"] = aggregate_frame_tokens(obj["frame_tokens"])
    return out


def dedup_captions(captions):
    """run_video_CapFilt.py:185-188: order-preserving exact-match dedup."""
    out = []
    for c in captions:
        if c not in out:
            out.append(c)
    return out


def keep_caption(itm_scores, threshold, mode="max_filter"):
    """run_video_CapFilt.py:116-123."""
    s = np.asarray(itm_scores, dtype=np.float32)
    prob = np.sum(s) / len(s) if mode == "avg_filter" else np.max(s)
    return bool(prob > threshold)


def shard_bounds(n, world, rank):
    """The reference's contiguous sharding, run_video_CapFilt.py:237-241 /
    run_visual_tokenization.py:427-431: step = n // world + 1."""
    step = n // world + 1
    start = rank * step
    return start, min(n, start + step)
