"""oracle/beam_ref.py against the beam search of the installed ``transformers`` (executable third-party code).

rule="5.15" must reproduce ``generate(num_beams=...)`` exactly; rule="4.15" (the product's rule, the version
models/med.py:7-8 names) shares every mechanism except the three ``# RULE`` sites, and must agree with both
whenever no hypothesis is banked before max_length (no EOS in reach) — the regime of the random-weight benchmark."""
import numpy as np
import pytest

from oracle import beam_ref, hf_beam

pytest.importorskip("transformers")

V, EOS, PAD = 13, 2, 7      # (a zero pad id makes 5.15 fill its output with EOS: `pad_token_id or eos_token_id[0]`)
PROMPT = [5, 1, 3, 1]


def _run(rule, fn, B, nb, max_len, min_len, penalty=1.0):
    prompts = np.array([PROMPT] * B, dtype=np.int64)
    prompts[:, 1] = np.array([1, 4, 6])[np.arange(B) % 3]   # distinct contexts per image
    return prompts, beam_ref.beam_search(lambda ids, bi: fn(ids), prompts, num_beams=nb, max_length=max_len,
                                         min_length=min_len, eos_token_id=EOS, pad_token_id=PAD, rule=rule,
                                         repetition_penalty=penalty)


CONFIGS = [(3, 12, 6, 0.3), (3, 20, 5, 0.15), (2, 9, 5, 0.5), (4, 10, 7, 0.3), (6, 8, 5, 0.3), (3, 7, 5, 0.6),
           (5, 12, 5, 0.1)]


@pytest.mark.parametrize("nb,max_len,min_len,eos_boost", CONFIGS)
def test_rule_5_15_equals_installed_transformers(nb, max_len, min_len, eos_boost):
    n_cases = n_eos_ended = 0
    for seed in range(12):
        fn = hf_beam.table_logits_fn(V, seed, EOS, eos_boost=eos_boost, ban=(PAD,))
        prompts, (seqs, scores) = _run("5.15", fn, 3, nb, max_len, min_len)
        hseqs, hscores = hf_beam.hf_generate(fn, prompts, V, num_beams=nb, max_length=max_len, min_length=min_len,
                                             eos_token_id=EOS, pad_token_id=PAD)
        for b in range(3):
            assert seqs[b].tolist() == hseqs[b].tolist(), (seed, b, seqs[b], hseqs[b])
            assert scores[b] == pytest.approx(hscores[b], rel=1e-5, abs=1e-6)
            n_cases += 1
            n_eos_ended += int(seqs[b][-1] == EOS)
    assert n_cases == 36
    if eos_boost >= 0.3:
        assert n_eos_ended > 0         # the EOS paths (ban, rank rule, banking, early stop) were exercised


@pytest.mark.parametrize("penalty", [1.3, 2.0, 0.7])
@pytest.mark.parametrize("nb,max_len,min_len,eos_boost", [(3, 12, 6, 0.3), (3, 20, 5, 0.15), (2, 9, 5, 0.5)])
def test_repetition_penalty_equals_installed_transformers(nb, max_len, min_len, eos_boost, penalty):
    """``generate(..., repetition_penalty=p)`` with beam search (models/blip.py:154-161 passes its argument through): the
    processor acts on the log-probabilities of every token already in the row, prompt included, before MinLength."""
    changed = 0
    for seed in range(8):
        fn = hf_beam.table_logits_fn(V, 900 + seed, EOS, eos_boost=eos_boost, ban=(PAD,))
        prompts, (seqs, scores) = _run("5.15", fn, 3, nb, max_len, min_len, penalty)
        _, (plain, _) = _run("5.15", fn, 3, nb, max_len, min_len)
        hseqs, hscores = hf_beam.hf_generate(fn, prompts, V, num_beams=nb, max_length=max_len, min_length=min_len,
                                             eos_token_id=EOS, pad_token_id=PAD, repetition_penalty=penalty)
        for b in range(3):
            assert seqs[b].tolist() == hseqs[b].tolist(), (seed, b, seqs[b], hseqs[b])
            assert scores[b] == pytest.approx(hscores[b], rel=1e-5, abs=1e-6)
            changed += int(seqs[b].tolist() != plain[b].tolist())
    assert changed > 0                 # the penalty moved some searches (a vocabulary of 13 repeats tokens all the time)


@pytest.mark.parametrize("nb,max_len,min_len", [(3, 20, 5), (3, 12, 6), (2, 9, 5), (4, 10, 7)])
def test_rule_4_15_equals_5_15_and_transformers_when_nothing_ends_early(nb, max_len, min_len):
    for seed in range(8):
        fn = hf_beam.table_logits_fn(V, 100 + seed, EOS, eos_boost=0.0, ban=(PAD,))
        prompts, (s4, sc4) = _run("4.15", fn, 2, nb, max_len, min_len)
        _, (s5, sc5) = _run("5.15", fn, 2, nb, max_len, min_len)
        hseqs, hscores = hf_beam.hf_generate(fn, prompts, V, num_beams=nb, max_length=max_len, min_length=min_len,
                                             eos_token_id=EOS, pad_token_id=PAD)
        for b in range(2):
            assert len(s4[b]) == max_len
            assert s4[b].tolist() == s5[b].tolist() == hseqs[b].tolist()
            # same ranking, different normaliser: 4.15 divides by the full length, 5.15 by the generated length
            assert sc4[b] * max_len == pytest.approx(hscores[b] * (max_len - len(PROMPT)), rel=1e-5)


def test_rules_differ_only_through_the_three_rule_sites():
    """With EOS in reach the two rules may pick different winners; when they do, re-scoring the 4.15 winner under the
    5.15 normaliser (and vice versa) explains the flip — i.e. candidate generation was identical."""
    flips = same = 0
    for seed in range(40):
        fn = hf_beam.table_logits_fn(V, 500 + seed, EOS, eos_boost=0.4, ban=(PAD,))
        t4, t5 = [], []
        prompts = np.array([PROMPT], dtype=np.int64)
        s4, _ = beam_ref.beam_search(lambda ids, bi: fn(ids), prompts, num_beams=3, max_length=12, min_length=5,
                                     eos_token_id=EOS, pad_token_id=PAD, rule="4.15", trace=t4)
        s5, _ = beam_ref.beam_search(lambda ids, bi: fn(ids), prompts, num_beams=3, max_length=12, min_length=5,
                                     eos_token_id=EOS, pad_token_id=PAD, rule="5.15", trace=t5)
        # identical candidate streams for as long as both searches run (the rules only decide what is banked / when to stop)
        for a, c in zip(t4, t5):
            assert np.array_equal(a["cand_index"], c["cand_index"]) and np.array_equal(a["cand_scores"], c["cand_scores"])
        if s4[0].tolist() == s5[0].tolist():
            same += 1
        else:
            flips += 1
    assert same > 0 and same + flips == 40
