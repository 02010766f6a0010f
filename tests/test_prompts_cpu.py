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


def test_prompt_strings_match_the_reference_class():
    """100 construct_prompt calls of the reference's Prompt class (three list templates x caption / qa / vlep /
    vlep-multichoice x v2 / v3 aggregation x random add_* subsets, list and string original captions) replayed on
    vidil_amd.prompts.Prompt with the same seeds: same strings, same in-place shuffle of a list-valued caption."""
    import copy

    from vidil_amd.prompts import Prompt

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "prompt_strings_golden.json")))
    assert len(g["cases"]) == 100
    seen_tasks = set()
    for c in g["cases"]:
        obj = copy.deepcopy(c["object"])
        call = c["call"]
        p = Prompt("PREFIX\n", seed=c["seed"])
        out = p.construct_prompt("vid", obj, c["frame_captions"], c["config"], call["question"], call["answer"], call["asr"],
                                 call["vlep_example"])
        assert out == c["prompt"], (c["seed"], c["config"])
        assert obj["caption"] == c["caption_after"]
        seen_tasks.add((c["config"]["prompt_task"], call["vlep_example"] is not None, c["config"]["prompt_temporal_template"]))
    assert len(seen_tasks) == 12


def test_fixed_prefix_prompt_lines_match_the_reference_script():
    """generate_prompts_fixed_prefix.py's save_prompt_lines (run here on seeded inputs: caption and qa task, with and
    without the unfiltered-caption fallback, empty subtitles) vs fixed_prefix_prompt_lines: same JSONL lines, same
    line -> video map."""
    import copy

    from vidil_amd.prompts import Prompt, fixed_prefix_prompt_lines

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "prompt_strings_golden.json")))
    assert len(g["fixed_prefix_runs"]) == 4
    for r in g["fixed_prefix_runs"]:
        cfg = copy.deepcopy(r["config"])
        lines, idx = fixed_prefix_prompt_lines(copy.deepcopy(r["visual_tokens"]), r["filtered"], r["unfiltered"], Prompt("PRE ", seed=7),
                                               cfg, r["qa"], r["asr"])
        assert lines == r["lines"] and len(lines) >= 2
        assert json.loads(json.dumps(idx)) == r["idx"]            # JSON turns the int keys / tuple values into str / list


def test_list_templates_refuse_what_the_reference_tables_lack():
    import pytest

    from vidil_amd.prompts import render_list

    assert render_list(["a", "b"], "temporal_natural") == "First, a. Then, b."
    assert render_list(list("abcd"), "temporal_natural") == "First, a. Then, b. After that, c. Finally, d."
    assert render_list(list("abcde"), "temporal_natural") == "First, a. Then, b. Then, c. Then, d. Finally, e."
    assert render_list(["a"], "temporal_index") == "[1] a." and render_list(["a", "b"], "static") == "a. b."
    for bad in ([], list("abcdefghi")):
        with pytest.raises(KeyError):
            render_list(bad, "static")
    with pytest.raises(NotImplementedError):
        render_list(["a"], "bullet")


def test_random_prefix_example_selection_and_lines_match_the_reference_script():
    """generate_prompts_random_prefix.py run here on seeded inputs (tests/golden/make_prompts_golden.py): the few-shot
    prefix — videos drawn with the global `random` seeded as the script seeds it, ground truth filled in, optional shuffled
    permutations — its `__chosen_samples.json`, and the query lines written with that prefix (vlep subtitles trimmed and
    capped at 1,024 characters, empty subtitle lists, videos without annotation skipped) vs vidil_amd.prompts."""
    import copy

    from vidil_amd.prompts import Prompt, random_prefix_examples, random_prefix_prompt_lines

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "prompt_strings_golden.json")))
    runs = g["random_prefix_runs"]
    assert len(runs) == 5 and {r["config"]["prompt_task"] for r in runs} == {"caption", "qa", "vlep"}
    for r in runs:
        cfg = copy.deepcopy(r["config"])
        prefixes, chosen = random_prefix_examples(copy.deepcopy(r["train_visual_tokens"]), r["train_filtered"], r["train_unfiltered"],
                                                  r["training_video_ids"], "INSTRUCTION LINE", cfg, r["qa"], r["asr"], shot=r["shot"], seed=r["seed"])
        assert prefixes == r["prefixes"], (r["config"]["prompt_task"], r["seed"])
        assert chosen == r["chosen"]
        cfg2 = dict(cfg, add_original_caption=False, add_answer=False)
        lines, idx = random_prefix_prompt_lines(copy.deepcopy(r["visual_tokens"]), r["filtered"], r["unfiltered"], Prompt(prefixes[0], seed=r["seed"]),
                                                cfg2, r["test_qa"], r["asr"])
        assert lines == r["lines"]
        assert {str(k): (list(v) if isinstance(v, tuple) else v) for k, v in idx.items()} == r["idx"]
    assert any(len(r["prefixes"]) > 1 for r in runs)          # the permutation branch ran
