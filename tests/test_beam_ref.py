"""Known-answer tests pinning oracle/beam_ref.py (the restated HF 4.15 beam search) on
hand-built logit tables whose outcome is derived by hand in tests/beam_cases.py."""
import math

import numpy as np
import pytest

import beam_cases as bc
from oracle import beam_ref


def run_case(case, B=1):
    prompts = np.array([bc.PROMPT] * B, dtype=np.int64)
    trace = []
    seqs, scores = beam_ref.beam_search(lambda ids, bi: bc.table_logits(case["table"], ids), prompts,
                                        num_beams=case["num_beams"], max_length=case["max_length"],
                                        min_length=case["min_length"], eos_token_id=bc.EOS, pad_token_id=bc.PAD, trace=trace)
    return seqs, scores, trace


def test_case_a_eos_ban_rank_rule_length_normalisation_and_done():
    seqs, scores, trace = run_case(bc.CASE_A)
    assert seqs[0].tolist() == bc.CASE_A["expect_tokens"]
    assert scores[0] == pytest.approx(bc.CASE_A["expect_score"], rel=1e-6)
    # step 0: EOS (p=0.6) is banned because cur_len 4 < min_length 5 -> candidates are b, c
    assert (trace[0]["cand_index"][0][:2] % bc.V).tolist() == [bc.Bt, bc.Ct]
    assert (trace[0]["cand_index"][0][:2] // bc.V).tolist() == [0, 0]      # both from beam 0 (others start at -1e9)
    # three forward passes only: lengths 4, 5, 6 (stops at max_length 7)
    assert [t["cur_len"] for t in trace] == [4, 5, 6]


def test_case_b_finalize_from_running_beams_no_eos_appended():
    seqs, scores, _ = run_case(bc.CASE_B)
    assert seqs[0].tolist() == bc.CASE_B["expect_tokens"]
    assert len(seqs[0]) == bc.CASE_B["max_length"]
    assert scores[0] == pytest.approx(bc.CASE_B["expect_score"], rel=1e-6)


def test_batch_rows_are_independent():
    s1, sc1, _ = run_case(bc.CASE_C, B=1)
    s3, sc3, _ = run_case(bc.CASE_C, B=3)
    for b in range(3):
        assert s3[b].tolist() == s1[0].tolist()
        assert sc3[b] == sc1[0]


def test_hypothesis_heap_keeps_best_and_tracks_worst():
    h = beam_ref.BeamHypotheses(2)
    h.add([1, 2, 3, 4], -4.0)          # -1.0
    h.add([1, 2, 3, 4, 5], -2.5)       # -0.5
    assert h.worst_score == -1.0
    h.add([1, 2], -1.0)                # -0.5 > worst -> replaces the -1.0 entry
    assert sorted(s for s, _ in h.beams) == [-0.5, -0.5] and h.worst_score == -0.5
    h.add([1, 2, 3], -3.0)             # -1.0 not > worst -> ignored
    assert len(h) == 2
    assert h.is_done(-2.0, 4) is True  # -0.5 >= -2/4
    assert h.is_done(-1.0, 4) is False
