"""Visual-token aggregation for the prompt generators — the consumer side of ``visual_tokens.json``
(visual_token_generation/prompts.py:52-118, used by generate_prompts_*.py).  CPU string work on the hot path's
output; no kernels.  Both functions reproduce the reference's tie behaviour (Python's stable sort over dict
insertion order), which fixes the result when counts are equal.

    v2: per category, count the top-2 tokens of every frame, keep the ``topk`` most frequent (first seen wins ties),
        then order them by the mean index of the frames they appeared in (temporal order).
    v3: split the frames into ``topk`` equal blocks; per block and category join the two most frequent of the top-2
        tokens (ties: lower summed rank, then first seen) with ', '; drop a block when it repeats the previously
        kept one; the LAST block is never emitted (the reference's loop stops one short).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

CATEGORIES = ("objects", "attributes", "scenes", "verbs")
FRAME_CANDIDATES = 2


def top_visual_tokens_v2(visual_tokens_object, topk):
    frame_tokens = visual_tokens_object["frame_tokens"]
    out = {}
    for key in CATEGORIES:
        stats = OrderedDict()                      # text -> [count, sum of frame indices], in first-seen order
        for i, ft in enumerate(frame_tokens):
            for s in ft.get(key, [])[:FRAME_CANDIDATES]:
                c = stats.setdefault(s, [0, 0])
                c[0] += 1
                c[1] += i
        ranked = sorted(stats.items(), key=lambda kv: kv[1][0], reverse=True)[:topk]      # stable: first seen wins ties
        ranked = sorted(ranked, key=lambda kv: kv[1][1] / kv[1][0])                        # temporal order
        out[key] = [text.rstrip(".") for text, _ in ranked]
    return out


def top_visual_tokens_v3(visual_tokens_object, topk):
    frame_tokens = visual_tokens_object["frame_tokens"]
    n = len(frame_tokens)
    starts = np.linspace(0, n, num=topk, dtype=int, endpoint=False)
    blocks = [(int(starts[i]), int(starts[i + 1]) if i + 1 < len(starts) else n) for i in range(len(starts))]
    out = {}
    for key in CATEGORIES:
        chosen = []
        for lo, hi in blocks:
            stats = OrderedDict()                  # text -> [count, summed rank]
            for i in range(lo, hi):
                for r in range(FRAME_CANDIDATES):
                    c = stats.setdefault(frame_tokens[i][key][r], [0, 0])
                    c[0] += 1
                    c[1] += r
            ranked = sorted(stats.items(), key=lambda kv: (-kv[1][0], kv[1][1]))
            chosen.append(", ".join(text.rstrip(".").strip() for text, _ in ranked[:FRAME_CANDIDATES]))
        keep = []
        for i in range(len(chosen) - 1):           # (sic) the reference never looks at the last block
            if i == 0 or chosen[i] != chosen[keep[-1]]:
                keep.append(i)
        out[key] = [chosen[i] for i in keep]
    return out
