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


def prompt_string_cases():
    """construct_prompt (visual_token_generation/prompts.py:120-313) over the template families, tasks and add_* flags,
    and the fixed-prefix generator's loop (generate_prompts_fixed_prefix.py:16-91; its module imports ruamel.yaml, which
    this image lacks and the function never uses: an empty stand-in module satisfies the import)."""
    import copy
    import itertools
    import tempfile
    import types

    rng = random.Random(99)
    cases = []
    base = dict(topk=4, visual_token_aggregation_version="v2", prompt_temporal_template="temporal_natural", prompt_task="caption",
                add_objects=True, add_events=True, add_attributes=True, add_scenes=True, add_ASR=True, add_original_caption=True,
                add_frame_captions=True, add_answer=True)
    grid = []
    for template, task in itertools.product(("temporal_natural", "temporal_index", "static"), ("caption", "qa", "vlep", "multichoice")):
        for version, topk in (("v2", 4), ("v3", 4), ("v2", 2), ("v2", 8), ("v3", 3)):
            grid.append(dict(base, prompt_temporal_template=template, prompt_task=task, visual_token_aggregation_version=version, topk=topk))
    for i in range(40):                                   # random flag subsets
        c = dict(rng.choice(grid))
        for k in [k for k in c if k.startswith("add_")]:
            c[k] = rng.random() < 0.6
        grid.append(c)
    for n, cfg in enumerate(grid):
        seed = 100 + n
        obj = synth(seed, rng.choice([4, 8, 8, 16]), rng.choice([2, 3, 6, 12]))
        obj["caption"] = rng.choice([["a man cooks. ", " two dogs run", "x y z"], " a single caption string ", [], ["only one"]])
        n_caps = rng.choice([1, 2, 3, 4, 5, 8, 9])
        fc = {"vid": [rng.choice(["a dog runs.", " a man talks . ", "cars on a road", "people dance.."]) + str(j) * (j % 2) for j in range(n_caps)]}
        task = cfg["prompt_task"]
        call = dict(question="what is shown?" if task == "qa" else None, answer="a dog" if task == "qa" else None,
                    asr=rng.choice([None, "hello there", ""]),
                    vlep_example=dict(events=["he leaves", "she stays"], answer=" A ") if task == "multichoice" else None)
        conf = dict(cfg, prompt_task="vlep" if task == "multichoice" else task)
        want_obj = copy.deepcopy(obj)
        p = Prompt("PREFIX\n", seed=seed)
        try:
            out = p.construct_prompt("vid", want_obj, fc, conf, call["question"], call["answer"], call["asr"], call["vlep_example"])
            err = None
        except Exception as e:                             # table lookups fail on 0 / > 8 items: the error type is the contract
            out, err = None, type(e).__name__
        cases.append(dict(seed=seed, object=obj, frame_captions=fc, config=conf, call=call, prompt=out, error=err,
                          caption_after=want_obj["caption"]))
    # the fixed-prefix generator's loop
    sys.modules.setdefault("ruamel", types.ModuleType("ruamel"))
    sys.modules.setdefault("ruamel.yaml", types.ModuleType("ruamel.yaml"))
    sys.modules["ruamel"].yaml = sys.modules["ruamel.yaml"]
    import generate_prompts_fixed_prefix as gp

    runs = []
    for task in ("caption", "qa"):
        for caption_all in (True, False):
            vt = {f"v{i}": synth(300 + i, 8, 5) for i in range(6)}
            filt = {f"v{i}": [f"cap {i} {j}." for j in range(1 + i)] for i in (0, 1, 3)}
            unf = {f"v{i}": [f"raw {i} {j}" for j in range(5)] for i in (0, 1, 2, 3, 4)}
            qa = {"v0": [dict(question="q0?", answer="a0"), dict(question="q1?", answer="a1")], "v2": [dict(question="q2?", answer="a2")]}
            asr = {"v0": ["hi", "there"], "v1": [], "v2": [""], "v3": ["x"]}
            with tempfile.TemporaryDirectory() as d:
                cfg = dict(base, prompt_task=task, add_events=False, add_scenes=False, add_original_caption=False, add_answer=False,
                           caption_all_video=caption_all, output_path=os.path.join(d, "out_q.jsonl"),
                           request_body=dict(engine="text-davinci-002", prompt="", temperature=0.0, max_tokens=64, top_p=1,
                                             frequency_penalty=0, presence_penalty=0))
                gp.save_prompt_lines(copy.deepcopy(vt), filt, unf, Prompt("PRE ", seed=7), cfg, qa if task == "qa" else None, asr)
                lines = open(cfg["output_path"]).read().splitlines()
                idx = json.load(open(os.path.join(d, "out_q__idx_2_videoid.json")))
            cfg_out = {k: v for k, v in cfg.items() if k != "output_path"}
            cfg_out["request_body"] = dict(cfg_out["request_body"], prompt="")
            runs.append(dict(visual_tokens=vt, filtered=filt, unfiltered=unf, qa=qa if task == "qa" else None, asr=asr, config=cfg_out,
                             lines=lines, idx=idx))
    # the random-prefix generator: few-shot example selection (get_prompt_prefix) + its own save_prompt_lines
    import contextlib
    import io

    import generate_prompts_random_prefix as gr

    rruns = []
    for task, permutate, shot, seed in (("caption", -1, 3, 42), ("qa", -1, 2, 7), ("vlep", -1, 3, 5), ("caption", 2, 3, 11), ("qa", 3, 3, 1)):
        tvt = {f"t{i}": synth(500 + i, 8, 5) for i in range(9) if i != 4}         # t4: an id without visual tokens
        for i, o in enumerate(tvt.values()):
            o["caption"] = [f"gt caption {i} a.", f"gt caption {i} b"] if i % 2 else f" single gt {i} "
        ids = sorted(f"t{i}" for i in range(9))
        tf = {f"t{i}": [f"train cap {i} {j}." for j in range(1 + i % 3)] for i in (0, 1, 2, 3, 5, 7)}
        tu = {f"t{i}": [f"train raw {i} {j}" for j in range(4)] for i in (0, 1, 2, 3, 5, 6, 7)}
        # (with permutations requested every chosen video must yield an example — the script indexes the shuffled
        #  permutations blindly — so only the plain runs exercise the "skip a video without annotation" path)
        qa = {f"t{i}": [dict(question=f"tq{i}-{j}?", answer=f"ta{i}-{j}") for j in range(1 + i % 2)]
              for i in ((0, 1, 2, 3, 5, 6, 8) if permutate == -1 else range(9))}
        if permutate != -1:
            tf = {f"t{i}": [f"train cap {i} {j}." for j in range(1 + i % 3)] for i in range(9)}
        long_line = "word " * 60
        asr = {"t0": ["hi", "there "], "t1": [], "t2": [" so, ", "what?", long_line, long_line, long_line, long_line, "late"], "t5": ["x"],
               "v0": ["test hi"], "v1": []}
        vt = {f"v{i}": synth(700 + i, 8, 4) for i in range(4)}
        filt = {f"v{i}": [f"cap {i} {j}." for j in range(2 + i)] for i in (0, 1, 3)}
        unf = {f"v{i}": [f"raw {i} {j}" for j in range(5)] for i in range(4)}
        tqa = {"v0": [dict(question="q0?", answer="a0")], "v1": [dict(question="q1?", answer="a1"), dict(question="q1b?", answer="a1b")]}
        with tempfile.TemporaryDirectory() as d:
            cfg = dict(base, prompt_task=task, add_events=False, add_scenes=False, caption_all_video=True, permutate=permutate,
                       output_path=os.path.join(d, "out_q.jsonl"),
                       request_body=dict(engine="text-davinci-002", prompt="", n=1, temperature=0.0, max_tokens=64, top_p=1,
                                         frequency_penalty=0, presence_penalty=0))
            cfg_in = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items() if k != "output_path"}
            with contextlib.redirect_stdout(io.StringIO()):
                prefixes = gr.get_prompt_prefix(copy.deepcopy(tvt), tf, tu, ids, "INSTRUCTION LINE", cfg, qa if task == "qa" else None, asr, shot, seed)
            chosen = json.load(open(os.path.join(d, "out_q__chosen_samples.json")))
            # the script then writes the queries with the user's flags and each prefix
            cfg2 = dict(cfg, add_original_caption=False, add_answer=False)
            with contextlib.redirect_stdout(io.StringIO()):
                gr.save_prompt_lines(copy.deepcopy(vt), filt, unf, Prompt(prefixes[0], seed=seed), cfg2, tqa if task == "qa" else None, asr)
            lines = open(cfg["output_path"]).read().splitlines()
            idx = json.load(open(os.path.join(d, "out_q__idx_2_videoid.json")))
        rruns.append(dict(train_visual_tokens=tvt, train_filtered=tf, train_unfiltered=tu, training_video_ids=ids, qa=qa if task == "qa" else None,
                          asr=asr, config=cfg_in, shot=shot, seed=seed, prefixes=prefixes, chosen=chosen,
                          visual_tokens=vt, filtered=filt, unfiltered=unf, test_qa=tqa if task == "qa" else None, lines=lines, idx=idx))
    json.dump(dict(cases=cases, fixed_prefix_runs=runs, random_prefix_runs=rruns), open(os.path.join(HERE, "prompt_strings_golden.json"), "w"), indent=0)
    print("wrote", len(rruns), "random-prefix runs")
    print("wrote", len(cases), "construct_prompt cases,", len(runs), "fixed-prefix runs;",
          sum(c["error"] is not None for c in cases), "of the cases are errors")


def main():
    prompt_string_cases()
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
