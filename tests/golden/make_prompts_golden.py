"""Generates tests/golden/prompts_golden.json by running the reference's own aggregation code
(visual_token_generation/prompts.py:52-118, importable as is: stdlib + numpy) on seeded synthetic frame tokens."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from visual_token_generation.prompts import Prompt  # noqa: E402

CATS = ("objects", "attributes", "scenes", "verbs")


def synth(seed, n_frames, vocab):
    rng = random.Random(seed)
    frames = []
    for _ in range(n_frames):
        frames.append({k: [rng.choice([f"{k[:3]}{j}" + ("." if j % 4 == 0 else "") for j in range(vocab)]) for _ in range(5)]
                       for k in CATS})
    return {"frame_tokens": frames, "caption": ["c"]}


def main():
    p = Prompt("{x}")
    cases = []
    for seed, (n_frames, vocab, topk) in enumerate([(8, 3, 4), (8, 6, 4), (8, 2, 8), (16, 5, 4), (4, 3, 2), (8, 12, 3), (5, 4, 4)]):
        obj = synth(seed, n_frames, vocab)
        cases.append(dict(object=obj, topk=topk, v2=p._get_top_visual_tokens_v2("v", obj, topk),
                          v3=p._get_top_visual_tokens_v3("v", obj, topk)))
    json.dump(cases, open(os.path.join(HERE, "prompts_golden.json"), "w"), indent=0)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
