"""Hand-built logit tables for the beam-search known-answer tests (CPU oracle and device kernels)."""
import math

import numpy as np

V = 6
PAD, A, EOS, Bt, Ct, D = 0, 1, 2, 3, 4, 5
PROMPT = [D, A, Bt, A]
TINY = 1e-9


def _row(p):
    r = np.full(V, math.log(TINY), dtype=np.float32)
    for t, v in p.items():
        r[t] = math.log(v)
    return r


def table_logits(table, ids):
    """logits[row] = table[(cur_len, last_token)] (log-probabilities, so log_softmax is ~identity)."""
    cur_len = ids.shape[1]
    return np.stack([_row(table[(cur_len, int(t))]) for t in ids[:, -1]])


# Scenario A: EOS banned before min_length; EOS beyond rank<num_beams ignored; hypothesis length counts
# the prompt and not the EOS; `done` fires when worst kept >= best running / cur_len.
CASE_A = dict(
    table={(4, A): {EOS: 0.6, Bt: 0.3, Ct: 0.1},
           (5, Bt): {EOS: 0.55, A: 0.45}, (5, Ct): {EOS: 0.9, D: 0.1},
           (6, A): {EOS: 0.8, Bt: 0.2}, (6, D): {EOS: 0.55, Ct: 0.45},
           (5, PAD): {A: 1.0}, (6, PAD): {A: 1.0}},
    num_beams=2, max_length=7, min_length=5,
    expect_tokens=[D, A, Bt, A, Bt, EOS],
    expect_score=(math.log(0.3) + math.log(0.55)) / 5.0,
)

# Scenario B: nothing ever ends; finalize ranks the running beams by score / max_length and appends no EOS.
CASE_B = dict(
    table={(4, A): {Bt: 0.6, Ct: 0.4},
           (5, Bt): {Bt: 0.6, Ct: 0.4}, (5, Ct): {Bt: 0.7, Ct: 0.3}},
    num_beams=2, max_length=6, min_length=5,
    expect_tokens=[D, A, Bt, A, Bt, Bt],
    expect_score=(math.log(0.6) + math.log(0.6)) / 6.0,
)

# Scenario C: three beams, an early short hypothesis is later displaced from the heap by better (longer) ones.
CASE_C = dict(
    table={(4, A): {Bt: 0.5, Ct: 0.3, D: 0.2},
           (5, Bt): {EOS: 0.05, A: 0.9, D: 0.05}, (5, Ct): {EOS: 0.4, A: 0.6}, (5, D): {EOS: 0.3, A: 0.7},
           (6, A): {EOS: 0.95, Bt: 0.05}, (6, D): {EOS: 0.5, A: 0.5},
           (7, Bt): {EOS: 0.9, A: 0.1}, (7, A): {EOS: 0.9, Bt: 0.1}},
    num_beams=3, max_length=8, min_length=5,
    expect_tokens=None, expect_score=None,   # oracle vs device only
)
