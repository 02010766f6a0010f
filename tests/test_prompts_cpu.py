"""Token aggregation of the prompt generators (vidil_amd/prompts.py) against golden outputs of the reference's own
functions (tests/golden/make_prompts_golden.py ran visual_token_generation/prompts.py:52-118 on seeded inputs)."""
import json
import os

from common import ROOT
from vidil_amd.prompts import top_visual_tokens_v2, top_visual_tokens_v3


def test_v2_and_v3_match_reference_goldens():
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "prompts_golden.json")))
    assert len(cases) == 7
    for c in cases:
        assert top_visual_tokens_v2(c["object"], c["topk"]) == c["v2"]
        assert top_visual_tokens_v3(c["object"], c["topk"]) == c["v3"]


def test_v3_never_emits_the_last_block_and_merges_repeats():
    frames = [{k: ["a", "b", "x", "y", "z"] for k in ("objects", "attributes", "scenes", "verbs")} for _ in range(8)]
    out = top_visual_tokens_v3({"frame_tokens": frames}, 4)
    assert out["objects"] == ["a, b"]               # four identical blocks -> one entry (and the last is skipped anyway)
