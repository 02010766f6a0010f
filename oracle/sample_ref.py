"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of one nucleus-sampling step of the caption
decoder and of the sampling loop.

What the reference runs (third party, not under /root/reference): HF ``transformers`` 4.15 ``GenerationMixin.sample``
as configured by models/blip.py:140-151 / run_video_CapFilt.py:103-104::

    text_decoder.generate(input_ids, max_length, min_length, do_sample=True, top_p=0.9, num_return_sequences=1,
                          eos_token_id=[SEP], pad_token_id=[PAD], repetition_penalty=1.1, **encoder kwargs)

with BertConfig defaults top_k=50, temperature=1.0.  Per step (4.15 ``_get_logits_processor`` / ``_get_logits_warper``
order): RepetitionPenaltyLogitsProcessor -> MinLengthLogitsProcessor -> TopKLogitsWarper(50) -> TopPLogitsWarper(0.9)
-> softmax -> torch.multinomial -> finished rows emit pad -> a drawn eos finishes the row.

PARITY STATUS: the logits processing / warping above is restated from the published 4.15 algorithm (unpinned against
executable reference code, like the beam search: transformers 4.15 is not installable here).  The random draw itself
CANNOT match the reference — torch.multinomial consumes the CUDA generator's Philox stream in an implementation-defined
way — so this build defines its own contract, shared by this file and csrc/sample.hip: u = Philox4x32-10(key = seed,
counter = (row, step, 0, 0))[0] >> 8 scaled to [0, 1), inverse-CDF over the surviving candidates ordered by (score
descending, token id ascending).  Samples are then a deterministic function of (logits, sequence, seed, row, step) and
are distributed exactly as the reference's.
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """Salmon et al., Philox4x32 with 10 rounds.  counter: 4 uint32, key: 2 uint32 -> 4 uint32."""
    c = [int(x) & MASK for x in counter]
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c


def uniform(seed, row, step):
    x = philox4x32_10([row, step, 0, 0], [seed & MASK, (seed >> 32) & MASK])[0]
    return np.float32(x >> 8) * np.float32(1.0 / 16777216.0)


def warp_row(logits, seq, cur_len, *, min_length, eos, top_k=50, top_p=0.9, rep_penalty=1.1, max_cand=64):
    """Steps 1-4 for one row.  Returns (candidate token ids, candidate scores f32) in draw order, after the
    nucleus cut, plus the running f32 sums used for the cut and the draw."""
    s = np.asarray(logits, dtype=np.float32).copy()
    orig = np.asarray(logits, dtype=np.float32)
    p = np.float32(rep_penalty)
    for t in set(int(x) for x in seq[:cur_len]):
        s[t] = orig[t] * p if orig[t] < 0 else orig[t] / p
    if cur_len < min_length:
        s[eos] = -np.inf
    order = np.lexsort((np.arange(s.size), -s.astype(np.float64)))      # score desc, id asc
    kth = s[order[top_k - 1]]
    cand = [int(i) for i in order[:max_cand] if s[i] >= kth and s[i] > -np.inf]
    v = s[cand]
    run, c = np.float32(0), []
    for x in v:
        run = np.float32(run + np.float32(np.exp(np.float64(x - v[0]))))
        c.append(run)
    cut = np.float32(np.float32(top_p) * run)
    kept = len(cand)
    for i, ci in enumerate(c):
        if ci > cut:
            kept = i + 1
            break
    return cand[:kept], v[:kept], np.array(c[:kept], dtype=np.float32)


def sample_row(logits, seq, cur_len, *, seed, row, step, **kw):
    """Returns (token, margin): margin = distance of the draw from the nearest CDF boundary relative to the total
    mass (tests accept either neighbour when the device's exp differs in the last ulp and the margin is tiny)."""
    cand, _, c = warp_row(logits, seq, cur_len, **kw)
    u = uniform(seed, row, step)
    r = np.float32(u * c[-1])
    pick = len(cand) - 1
    for i, ci in enumerate(c):
        if ci > r:
            pick = i
            break
    margin = float(np.min(np.abs(c.astype(np.float64) - float(r))) / float(c[-1]))
    return cand[pick], margin


def sample_search(step_fn, prompt_ids, *, max_length, min_length, eos_token_id, pad_token_id, seed, top_k=50, top_p=0.9,
                  rep_penalty=1.1, row_offset=0, trace=None):
    """The sampling loop (HF 4.15 ``sample``): step_fn(ids [B,T]) -> logits f32 [B,V] of the last position.
    Returns ids [B,max_length] padded with pad_token_id (a drawn eos is part of the sequence)."""
    ids = np.asarray(prompt_ids, dtype=np.int64)
    B = ids.shape[0]
    out = np.full((B, max_length), pad_token_id, dtype=np.int64)
    out[:, :ids.shape[1]] = ids
    done = np.zeros(B, dtype=bool)
    cur_len, step = ids.shape[1], 0
    while cur_len < max_length and not done.all():
        logits = step_fn(out[:, :cur_len])
        for b in range(B):
            if done[b]:
                out[b, cur_len] = pad_token_id
                continue
            tok, margin = sample_row(logits[b], out[b], cur_len, seed=seed, row=row_offset + b, step=step, min_length=min_length,
                                     eos=eos_token_id, top_k=top_k, top_p=top_p, rep_penalty=rep_penalty)
            if trace is not None:
                trace.append(dict(step=step, row=b, token=tok, margin=margin))
            out[b, cur_len] = tok
            if tok == eos_token_id:
                done[b] = True
        cur_len += 1
        step += 1
    return out
