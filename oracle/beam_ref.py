"""oracle/beam_ref.py — TEST INFRASTRUCTURE.  Restatement of the beam search the
reference gets from HuggingFace ``transformers`` (call site models/blip.py:154-161:
``text_decoder.generate(num_beams=3, max_length=20, min_length=5, eos=[SEP],
pad=[PAD], repetition_penalty=1.0)``).

The algorithm lives in the third-party ``transformers`` (unpinned in
docker/requirements.txt:9; models/med.py:7-8 names v4.15.0), which is absent from
/root/reference and not installable here.  This file restates the published
v4.15.0 algorithm — ``generation_utils.GenerationMixin.beam_search``,
``generation_beam_search.BeamSearchScorer/BeamHypotheses`` and
``generation_logits_process.MinLengthLogitsProcessor`` — with
length_penalty=1.0, early_stopping=False, num_beam_groups=1,
num_return_sequences=1.

Pinning (tests/test_beam_ref.py, tests/test_beam_hf.py):
  * ``rule="5.15"`` switches the THREE places where the installed ``transformers``
    5.15 differs from 4.15 (each marked ``# RULE`` below): the hypothesis-score
    normaliser, the "no improvement possible" test, and how the last step is banked.
    In that mode the restatement is checked against the executable
    ``GenerationMixin.generate(num_beams=...)`` of the installed 5.15 on hundreds of
    seeded table language models (sequences identical, scores to 1e-6).  Everything
    else — log-softmax, the min-length EOS ban, the [0,-1e9,...] initial beam scores,
    the 2*num_beams candidates, the rank < num_beams rule for EOS, the choice of the
    next beams, beam_idx, the best-num_beams hypothesis set, the output format — is
    shared code between the two modes and therefore pinned by executable reference
    code.
  * the three 4.15 deltas remain hand-derived from the published 4.15 source and
    are pinned by the known-answer tables of tests/beam_cases.py; whenever no
    hypothesis ends with EOS before max_length (the case for the benchmark's
    random-init weights) the two rules rank identically, and the tests check
    4.15 == 5.15 == HF there.
"""
from __future__ import annotations

import numpy as np
import torch


class BeamHypotheses:
    """transformers 4.15 generation_beam_search.BeamHypotheses (5.15 keeps the same best-num_beams set through a
    merged top-k, generation/utils.py `_update_finished_beams`)."""

    def __init__(self, num_beams, length_penalty=1.0, early_stopping=False):
        self.num_beams = num_beams
        self.length_penalty = length_penalty
        self.early_stopping = early_stopping
        self.beams = []            # list of (score, token list), insertion ordered
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs, norm_len=None):
        """norm_len: the length the score is normalised by (default len(hyp), the 4.15 rule)."""
        score = sum_logprobs / ((len(hyp) if norm_len is None else norm_len) ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, list(hyp)))
            if len(self) > self.num_beams:
                ranked = sorted((s, idx) for idx, (s, _) in enumerate(self.beams))
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


def log_softmax_rows(logits):
    """f32 log-softmax as torch computes it (x - max - log(sum(exp(x - max))))."""
    return torch.log_softmax(torch.as_tensor(logits, dtype=torch.float32), dim=-1).numpy()


def beam_search(step_fn, prompt_ids, *, num_beams=3, max_length=20, min_length=5, eos_token_id=102,
                pad_token_id=0, trace=None, rule="4.15", repetition_penalty=1.0):
    """Run beam search.

    step_fn(input_ids[np.int64, rows x cur_len], beam_idx or None) -> logits[rows, V] (f32) for
    the LAST position.  ``beam_idx`` (np.int64[rows]) is the row gather applied to
    the sequences since the previous call (the caller reorders its KV cache with
    it, models/med.py:951-955); None on the first call.

    repetition_penalty: ``models/blip.py:161`` hands its own argument (default 1.0, what ``run_video_CapFilt.py:101``
    leaves it at) to ``generate``; a value != 1.0 installs a RepetitionPenaltyLogitsProcessor FIRST in the processor list
    (before MinLength), and beam search runs the list on the LOG-PROBABILITIES: every token already in a row's
    ``input_ids`` (prompt included) gets ``s * penalty if s < 0 else s / penalty`` (f32).

    rule: "4.15" (the version the reference names — the product's behaviour) or "5.15" (the installed
    version, executable here: tests/test_beam_hf.py); the differences are the three ``# RULE`` sites.

    Returns (sequences: list of np.int64 arrays incl. prompt and a trailing EOS
    when shorter than max_length, scores: list of float).
    """
    if rule not in ("4.15", "5.15"):
        raise ValueError(rule)
    v5 = rule == "5.15"
    prompt_ids = np.asarray(prompt_ids, dtype=np.int64)
    B, P = prompt_ids.shape
    nb = num_beams
    input_ids = np.repeat(prompt_ids, nb, axis=0)            # _expand_inputs_for_generation
    beam_scores = np.zeros((B, nb), dtype=np.float32)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.reshape(-1)
    hyps = [BeamHypotheses(nb) for _ in range(B)]
    done = [False] * B
    beam_idx = None
    cur_len = input_ids.shape[1]
    while True:
        logits = np.asarray(step_fn(input_ids, beam_idx), dtype=np.float32)
        V = logits.shape[-1]
        scores = log_softmax_rows(logits)
        if repetition_penalty != 1.0:                          # RepetitionPenaltyLogitsProcessor (gather / where / scatter_)
            pen = np.float32(repetition_penalty)
            for r in range(scores.shape[0]):
                toks = np.unique(input_ids[r])
                sc = scores[r, toks]
                scores[r, toks] = np.where(sc < 0, sc * pen, sc / pen).astype(np.float32)
        if cur_len < min_length:                               # MinLengthLogitsProcessor
            scores[:, eos_token_id] = -np.inf
        scores = scores + beam_scores[:, None]
        flat = torch.from_numpy(scores.reshape(B, nb * V))
        top_s, top_i = torch.topk(flat, 2 * nb, dim=1, largest=True, sorted=True)
        top_s, top_i = top_s.numpy(), top_i.numpy()
        if trace is not None:
            trace.append(dict(cur_len=cur_len, logits=logits.copy(), cand_scores=top_s.copy(), cand_index=top_i.copy(),
                              beam_scores=beam_scores.copy()))
        next_indices = top_i // V
        next_tokens = top_i % V
        last_step = cur_len + 1 >= max_length
        # ---- BeamSearchScorer.process (4.15) / _get_running_beams_for_next_iteration + _update_finished_beams (5.15)
        nbs = np.zeros((B, nb), dtype=np.float32)
        nbt = np.zeros((B, nb), dtype=np.int64)
        nbi = np.zeros((B, nb), dtype=np.int64)
        for b in range(B):
            if done[b] and not v5:
                nbs[b, :] = 0
                nbt[b, :] = pad_token_id
                nbi[b, :] = 0
                continue
            # (5.15 keeps running the beams of an image whose hypotheses can no longer improve, and only stops
            #  banking them; 4.15 pads the image out.  Neither changes what is returned.)
            slot = 0
            for rank in range(2 * nb):
                tok, sc, idx = int(next_tokens[b, rank]), float(top_s[b, rank]), int(next_indices[b, rank])
                row = b * nb + idx
                # RULE (last step): 5.15 banks the top num_beams candidates of the step that reaches max_length
                # through its MaxLengthCriteria; 4.15 lets them become beams and banks those in finalize() below.
                ends = tok == eos_token_id or (v5 and last_step)
                if ends:
                    if rank >= nb:
                        continue
                    if v5:
                        # RULE (normaliser): 5.15 divides by the GENERATED length including this token
                        # (generation/utils.py:3182); the banked sequence includes the token.
                        if not done[b]:
                            hyps[b].add(input_ids[row].tolist() + [tok], sc, norm_len=cur_len + 1 - P)
                    else:
                        # 4.15 divides by the length of the sequence so far — prompt included, EOS excluded
                        hyps[b].add(input_ids[row].tolist(), sc)
                else:
                    nbs[b, slot], nbt[b, slot], nbi[b, slot] = sc, tok, row
                    slot += 1
                if slot == nb:
                    break
            assert slot == nb or (v5 and last_step)
            # RULE ("no improvement possible"): with all num_beams hypotheses banked,
            #   4.15: worst kept >= (best of this step's 2*num_beams candidates, EOS ones included) / cur_len
            #   5.15: not (best NEW running beam / (cur_len + 1 - P) > worst kept)      [_check_early_stop_heuristic]
            if v5:
                if len(hyps[b]) >= nb and not last_step:
                    done[b] = done[b] or not (float(nbs[b, 0]) / (cur_len + 1 - P) > hyps[b].worst_score)
            else:
                done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len)
        beam_scores = nbs.reshape(-1)
        beam_idx = nbi.reshape(-1)
        input_ids = np.concatenate([input_ids[beam_idx], nbt.reshape(-1, 1)], axis=1)
        cur_len += 1
        if all(done) or input_ids.shape[1] >= max_length:
            break
    # ---- BeamSearchScorer.finalize (4.15 only: 5.15 has banked everything inside the loop)
    if not v5:
        for b in range(B):
            if done[b]:
                continue
            for j in range(nb):
                row = b * nb + j
                hyps[b].add(input_ids[row].tolist(), float(beam_scores[row]))
    seqs, best_scores = [], []
    for b in range(B):
        ranked = sorted(hyps[b].beams, key=lambda x: x[0])
        s, toks = ranked.pop()
        toks = list(toks)
        if not v5 and len(toks) < max_length:
            toks.append(eos_token_id)
        seqs.append(np.asarray(toks, dtype=np.int64))
        best_scores.append(s)
    return seqs, best_scores
