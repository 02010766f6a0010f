"""CPU checks of the nucleus-sampling restatement (oracle/sample_ref.py): Philox known answers (Random123 vectors),
hand-derived known-answer cases for every logits processor / warper of HF 4.15's sample() as BLIP configures it,
and the distribution of the draws."""
import numpy as np

from oracle import sample_ref as S

EOS, PAD = 102, 0


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors: philox4x32 10
    assert S.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert S.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert S.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = [float(S.uniform(7, r, s)) for r in range(50) for s in range(20)]
    assert 0.0 <= min(u) and max(u) < 1.0 and abs(np.mean(u) - 0.5) < 0.03


def _logits_from_probs(probs, V=200, floor=-30.0):
    lg = np.full(V, floor, dtype=np.float32)
    lg[10:10 + len(probs)] = np.log(np.asarray(probs, dtype=np.float64)).astype(np.float32)
    return lg


def test_top_p_cut_keeps_the_token_that_crosses_the_threshold():
    lg = _logits_from_probs([0.5, 0.3, 0.15, 0.05])
    cand, v, c = S.warp_row(lg, [1, 2, 3, 4], 4, min_length=0, eos=EOS, top_k=50, top_p=0.9, rep_penalty=1.0)
    assert cand == [10, 11, 12]                       # cum = .5, .8, .95 -> the third crosses 0.9 and is kept
    cand, _, _ = S.warp_row(lg, [1, 2, 3, 4], 4, min_length=0, eos=EOS, top_k=50, top_p=0.79, rep_penalty=1.0)
    assert cand == [10, 11]
    cand, _, _ = S.warp_row(lg, [1, 2, 3, 4], 4, min_length=0, eos=EOS, top_k=50, top_p=0.3, rep_penalty=1.0)
    assert cand == [10]                               # at least one token always survives


def test_top_k_keeps_ties_with_the_kth_and_orders_by_id():
    lg = np.full(300, -20.0, dtype=np.float32)
    lg[[5, 9, 200]] = 3.0
    lg[[7, 8]] = 1.0                                   # k = 4: the 4th largest is 1.0, both 1.0s stay
    cand, _, _ = S.warp_row(lg, [1], 1, min_length=0, eos=EOS, top_k=4, top_p=1.0, rep_penalty=1.0)
    assert cand == [5, 9, 200, 7, 8]


def test_repetition_penalty_once_per_distinct_token_and_sign_rule():
    lg = np.zeros(50, dtype=np.float32)
    lg[3], lg[4], lg[5] = 2.2, -2.0, 1.0
    cand, v, _ = S.warp_row(lg, [3, 3, 4, 3], 4, min_length=0, eos=EOS, top_k=50, top_p=1.0, rep_penalty=1.1)
    d = dict(zip(cand, v))
    assert np.isclose(d[3], 2.2 / 1.1) and np.isclose(d[4], -2.0 * 1.1) and d[5] == 1.0


def test_min_length_bans_eos_only_while_short():
    lg = np.full(200, -5.0, dtype=np.float32)
    lg[EOS] = 9.0
    cand, _, _ = S.warp_row(lg, [1, 2, 3, 4], 4, min_length=5, eos=EOS, top_k=50, top_p=0.9, rep_penalty=1.0)
    assert EOS not in cand
    cand, _, _ = S.warp_row(lg, [1, 2, 3, 4, 5], 5, min_length=5, eos=EOS, top_k=50, top_p=0.9, rep_penalty=1.0)
    assert cand == [EOS]


def test_draws_follow_the_renormalised_nucleus_distribution():
    probs = [0.4, 0.25, 0.2, 0.1, 0.05]
    lg = _logits_from_probs(probs)
    counts = {}
    n = 6000
    for s in range(n):
        tok, _ = S.sample_row(lg, [1], 1, seed=1234, row=3, step=s, min_length=0, eos=EOS, top_k=50, top_p=0.9,
                              rep_penalty=1.0)
        counts[tok] = counts.get(tok, 0) + 1
    assert set(counts) == {10, 11, 12, 13}            # cum .4 .65 .85 .95: four survive, renormalised by .95
    for i, p in enumerate(probs[:4]):
        assert abs(counts[10 + i] / n - p / 0.95) < 0.02


def test_sampling_loop_pads_after_eos_and_stops_at_max_length():
    V = 300

    def step_fn(ids):
        lg = np.full((ids.shape[0], V), -30.0, dtype=np.float32)
        for b in range(ids.shape[0]):
            if b == 0 and ids.shape[1] >= 6:
                lg[b, EOS] = 20.0                      # row 0 ends as soon as min_length allows
            else:
                lg[b, 50 + ids.shape[1]] = 20.0
        return lg

    prompt = np.array([[250, 251, 252, 253]] * 2)
    out = S.sample_search(step_fn, prompt, max_length=10, min_length=5, eos_token_id=EOS, pad_token_id=PAD, seed=9)
    assert list(out[0]) == [250, 251, 252, 253, 54, 55, EOS, PAD, PAD, PAD]
    assert list(out[1]) == [250, 251, 252, 253, 54, 55, 56, 57, 58, 59]
