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


# ----------------------------------------------------------------------------------------------------------------
# Prompt text — what the generate_prompts_*.py scripts turn the hot path's three JSON files into
# (visual_token_generation/prompts.py:9-40,120-313; generate_prompts_fixed_prefix.py:16-91).  Pure string work.

def _numbered(items, word_of):
    return " ".join(f"{word_of(i, len(items))} {x}." for i, x in enumerate(items))


def _natural_word(i, n):
    """First / Then / ... / Finally as the reference's tables spell them: a 2-item list ends with 'Then,', only the
    4-item list says 'After that,' (visual_token_generation/prompts.py:9-18)."""
    if i == 0:
        return "First,"
    if i == n - 1 and n > 2:
        return "Finally,"
    if n == 4 and i == 2:
        return "After that,"
    return "Then,"


TEMPLATES = {
    "temporal_natural": lambda xs: _numbered(xs, _natural_word),
    "temporal_index": lambda xs: _numbered(xs, lambda i, n: f"[{i + 1}]"),
    "static": lambda xs: " ".join(f"{x}." for x in xs),
}
MAX_TEMPLATE_ITEMS = 8        # the reference's lookup tables end at 8 entries and have none for an empty list


def render_list(items, template):
    """One of the three list templates; like the reference's table lookup it refuses 0 or more than 8 items."""
    if template not in TEMPLATES:
        raise NotImplementedError(template)
    if not 1 <= len(items) <= MAX_TEMPLATE_ITEMS:
        raise KeyError(len(items))
    return TEMPLATES[template](items)


class Prompt:
    """``visual_token_generation.prompts.Prompt``: a prefix (text, or a path to a text file) plus ``construct_prompt``.
    Like the reference it seeds the GLOBAL ``random`` module on construction and shuffles a list-valued original
    caption in place before taking its first entry, so a replay on the same seed yields the same strings."""

    def __init__(self, template_txt, seed=42):
        import os
        import random

        random.seed(seed)
        self.template = open(template_txt).read() if os.path.exists(template_txt) else template_txt

    def construct_prompt(self, video_name, visual_tokens_object, frame_captions, config, question=None, answer=None,
                         asr=None, vlep_example=None):
        topk = config["topk"]
        version = config["visual_token_aggregation_version"]
        if version not in ("v2", "v3"):
            raise UnboundLocalError("visual_token_aggregation_version must be 'v2' or 'v3'")   # the reference's failure
        tokens = (top_visual_tokens_v2 if version == "v2" else top_visual_tokens_v3)(visual_tokens_object, topk)
        template = config["prompt_temporal_template"]
        if template not in TEMPLATES:
            raise NotImplementedError(template)
        parts = {
            "Scene": tokens["scenes"][0] if config["add_scenes"] else None,
            "Objects": render_list(tokens["objects"], template) if config["add_objects"] else None,
            "Events": render_list(tokens["verbs"], template) if config["add_events"] else None,
            "Attributes": render_list(tokens["attributes"], template) if config["add_attributes"] else None,
        }
        # (rendered even when add_frame_captions is off, like the reference: a video without captions fails either way)
        caps = render_list([c.rstrip(".").strip() for c in frame_captions[video_name][:topk]], template)
        parts["Frame Captions"] = caps if config["add_frame_captions"] else None
        dialogue = vlep_example is not None or config["prompt_task"] == "vlep"
        parts["Dialogue" if dialogue else "Subtitle"] = asr if config["add_ASR"] else None
        p = self.template + "".join(f"{k}: {v}\n" for k, v in parts.items() if v)
        if vlep_example is not None:
            a, b = vlep_example["events"]
            p += f"Question: What is more likely to happen next? A:{a} B:{b}\nAnswer:"
            if config["add_original_caption"]:
                p += " " + vlep_example["answer"].strip()
            return p
        original = visual_tokens_object["caption"] if config["add_original_caption"] else None
        task = config["prompt_task"]
        if task in ("caption", "vlep"):
            p += "Video Caption:" if task == "caption" else "What is likely to happen next?"
            if original:
                if isinstance(original, list):
                    import random

                    random.shuffle(original)
                    p += " " + original[0].strip()
                elif isinstance(original, str):
                    p += " " + original.strip()
        elif task == "qa":
            assert question is not None
            p += "Question: " + question + "\nAnswer:"
            if answer and config["add_answer"]:
                p += " " + answer
        return p


def fixed_prefix_prompt_lines(visual_tokens, frame_captions_filtered, frame_captions_unfiltered, prompt, config,
                              video_2_question_answer_pairs=None, video_2_asr=None):
    """The loop of generate_prompts_fixed_prefix.py:16-78 without its file output: one request body (JSON text) per
    video — per (video, question) for the qa task — and the line -> video map.  Videos without filtered captions fall
    back to the unfiltered ones when ``caption_all_video`` is set and are skipped otherwise; an empty subtitle list
    becomes 'no subtitle.'."""
    import json

    lines, line_to_video = [], {}
    for video_name, obj in visual_tokens.items():
        captions = frame_captions_filtered
        if video_name not in frame_captions_filtered:
            if not config["caption_all_video"] or video_name not in frame_captions_unfiltered:
                continue
            captions = frame_captions_unfiltered
        asr = None
        if video_2_asr is not None and video_name in video_2_asr:
            asr = " ".join(video_2_asr[video_name]) or "no subtitle."
        if config["prompt_task"] == "qa":
            if video_name not in video_2_question_answer_pairs:
                continue
            for qidx, item in enumerate(video_2_question_answer_pairs[video_name]):
                body = config["request_body"]
                body["prompt"] = prompt.construct_prompt(video_name, obj, captions, config, item["question"], item["answer"], asr)
                lines.append(json.dumps(body))
                line_to_video[len(lines) - 1] = (video_name, qidx)
        else:
            body = config["request_body"]
            body["prompt"] = prompt.construct_prompt(video_name, obj, captions, config, question=None, answer=None, asr=asr)
            lines.append(json.dumps(body))
            line_to_video[len(lines) - 1] = video_name
    return lines, line_to_video


def _subtitle_text(subs, task):
    """ASR lines -> one string (generate_prompts_random_prefix.py:60-81,143-164): an empty list becomes 'no subtitle.'; the
    vlep task strips every line, closes it with '.' unless it already ends in punctuation or a quote, and stops once
    1,024 characters are reached; the other tasks join the lines as they are."""
    if subs == []:
        return "no subtitle."
    if task != "vlep":
        return " ".join(subs)
    out, total = [], 0
    for sub in subs:
        sub = sub.strip()
        if not sub.endswith((".", ",", "?", ";", "!", ":", "'", '"')):
            sub += "."
        out.append(sub)
        total += len(sub)
        if total >= 1024:
            break
    return " ".join(out)


def random_prefix_examples(train_visual_tokens, train_frame_captions_filtered, train_frame_captions_unfiltered, training_video_ids,
                           instruction_line, config, video_2_question_answer_pairs=None, video_2_asr=None, shot=5, seed=42):
    """The few-shot prefix of generate_prompts_random_prefix.py:16-134 (``get_prompt_prefix``) without its file output:
    ``shot`` distinct training videos drawn with ``random.choice`` on the GLOBAL generator seeded with ``seed`` (after a
    throw-away ``Prompt("", seed)``, as the script does), each rendered by ``construct_prompt`` with the ground truth filled
    in by the caller's config (add_original_caption / add_answer), joined under the instruction line; ``permutate`` > 0
    yields that many shuffled orders of the examples.  Returns (list of prefix strings, chosen examples as the script's
    ``__chosen_samples.json`` holds them).  Videos whose captions are missing are skipped exactly like there."""
    import itertools
    import random

    dummy = Prompt("", seed=seed)
    random.seed(seed)
    chosen = []
    while len(chosen) != shot:
        cand = random.choice(training_video_ids)
        if cand in train_visual_tokens and cand not in chosen:
            chosen.append(cand)
    picked, examples = {}, []
    task = config["prompt_task"]
    for video_name in chosen:
        obj = train_visual_tokens[video_name]
        captions = train_frame_captions_filtered
        if video_name not in train_frame_captions_filtered:
            if not config["caption_all_video"] or video_name not in train_frame_captions_unfiltered:
                continue
            captions = train_frame_captions_unfiltered
        asr = None
        if video_2_asr is not None and video_name in video_2_asr:
            asr = _subtitle_text(video_2_asr[video_name], task)
        if task == "qa":
            if video_name not in video_2_question_answer_pairs:
                continue
            item = random.choice(video_2_question_answer_pairs[video_name])
            text = dummy.construct_prompt(video_name, obj, captions, config, item["question"], item["answer"], asr)
            picked[video_name] = {"question": item["question"], "answer": item["answer"]}
        elif task in ("caption", "vlep"):
            text = dummy.construct_prompt(video_name, obj, captions, config, question=None, answer=None, asr=asr)
            marker = "Video Caption:" if task == "caption" else "What is likely to happen next?"
            picked[video_name] = [text.split(marker)[-1].strip()]
        else:
            raise UnboundLocalError("prompt_task must be 'qa', 'caption' or 'vlep'")       # the reference's failure
        examples.append(text)
    if config["permutate"] == -1:
        return ["\n\n".join([instruction_line] + examples) + "\n\n"], picked
    orders = list(itertools.permutations(examples))
    random.shuffle(orders)
    return ["\n\n".join([instruction_line] + list(orders[i])) + "\n\n" for i in range(config["permutate"])], picked


def random_prefix_prompt_lines(visual_tokens, frame_captions_filtered, frame_captions_unfiltered, prompt, config,
                               video_2_question_answer_pairs=None, video_2_asr=None):
    """``save_prompt_lines`` of generate_prompts_random_prefix.py:136-210 without its file output: the loop of
    ``fixed_prefix_prompt_lines`` with that script's subtitle handling (vlep: trimmed and capped at 1,024 characters)."""
    import json

    lines, line_to_video = [], {}
    task = config["prompt_task"]
    for video_name, obj in visual_tokens.items():
        captions = frame_captions_filtered
        if video_name not in frame_captions_filtered:
            if not config["caption_all_video"] or video_name not in frame_captions_unfiltered:
                continue
            captions = frame_captions_unfiltered
        asr = None
        if video_2_asr is not None and video_name in video_2_asr:
            asr = _subtitle_text(video_2_asr[video_name], task)
        if task == "qa":
            if video_name not in video_2_question_answer_pairs:
                continue
            for qidx, item in enumerate(video_2_question_answer_pairs[video_name]):
                body = config["request_body"]
                body["prompt"] = prompt.construct_prompt(video_name, obj, captions, config, item["question"], item["answer"], asr)
                lines.append(json.dumps(body))
                line_to_video[len(lines) - 1] = (video_name, qidx)
        else:
            body = config["request_body"]
            body["prompt"] = prompt.construct_prompt(video_name, obj, captions, config, question=None, answer=None, asr=asr)
            lines.append(json.dumps(body))
            line_to_video[len(lines) - 1] = video_name
    return lines, line_to_video
