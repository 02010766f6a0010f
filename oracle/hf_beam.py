"""oracle/hf_beam.py — TEST INFRASTRUCTURE.  Drives the beam search of the INSTALLED HuggingFace
``transformers`` (5.x; ``GenerationMixin.generate(num_beams=...)``) on a table language model, so that
oracle/beam_ref.py (rule="5.15") and, through it, every mechanism it shares with the 4.15 rule the
product implements, is pinned by executable third-party code instead of by hand-derived answers alone.

The reference's call site is models/blip.py:154-161 (``self.text_decoder.generate(input_ids=...,
max_length, min_length, num_beams, eos_token_id=sep, pad_token_id=pad, repetition_penalty=1.0)``);
the reference's own ``BertLMHeadModel`` cannot run under 5.x (tuple caches, ``past=`` keyword), hence
the stub: ``logits_fn(ids[np.int64 rows x cur_len]) -> f32 [rows, V]`` is the whole language model.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


def table_logits_fn(V, seed, eos, eos_boost=0.3, eos_from=0, ban=(0,), scale=2.0):
    """A deterministic 'language model': logits are a seeded function of the whole sequence, so beams with
    different histories see different distributions.  ``eos_boost``: probability that EOS gets +3 (from
    length ``eos_from`` on); tokens in ``ban`` (the pad id) are never competitive."""
    cache = {}

    def fn(ids):
        out = np.empty((ids.shape[0], V), dtype=np.float32)
        for r, row in enumerate(np.asarray(ids).tolist()):
            key = tuple(int(t) for t in row)
            if key not in cache:
                h = hash((seed,) + key) & 0xFFFFFFFF
                rng = np.random.default_rng(h)
                l = (rng.standard_normal(V) * scale).astype(np.float32)
                if len(key) >= eos_from and rng.random() < eos_boost:
                    l[eos] += 3.0
                elif eos_boost == 0.0:
                    l[eos] = -30.0
                for t in ban:
                    l[t] = -30.0
                cache[key] = l
            out[r] = cache[key]
        return out

    return fn


def hf_generate(logits_fn, prompt_ids, V, *, num_beams, max_length, min_length, eos_token_id, pad_token_id,
                repetition_penalty=1.0):
    """Returns (list of np.int64 sequences with trailing pads stripped, list of float scores) from the installed
    transformers' beam search (length_penalty 1.0, early_stopping False, no sampling, no cache)."""
    from transformers import GenerationMixin, PretrainedConfig, PreTrainedModel
    from transformers.modeling_outputs import CausalLMOutput

    class _Cfg(PretrainedConfig):
        model_type = "vidil_table_lm"

        def __init__(self, vocab_size=8, **kw):
            super().__init__(**kw)
            self.vocab_size = vocab_size

    class _LM(PreTrainedModel, GenerationMixin):
        config_class = _Cfg

        def __init__(self, config):
            super().__init__(config)
            self.dummy = nn.Parameter(torch.zeros(1))

        def forward(self, input_ids=None, attention_mask=None, **kw):
            logits = torch.zeros(input_ids.shape[0], input_ids.shape[1], self.config.vocab_size)
            logits[:, -1] = torch.from_numpy(logits_fn(input_ids.numpy()))
            return CausalLMOutput(logits=logits)

    import transformers

    transformers.logging.set_verbosity_error()
    m = _LM(_Cfg(vocab_size=V))
    out = m.generate(torch.as_tensor(np.asarray(prompt_ids), dtype=torch.long), num_beams=num_beams, max_length=max_length,
                     min_length=min_length, eos_token_id=eos_token_id, pad_token_id=pad_token_id, do_sample=False,
                     use_cache=False, length_penalty=1.0, early_stopping=False, repetition_penalty=float(repetition_penalty),
                     return_dict_in_generate=True, output_scores=True)
    seqs = []
    for row in out.sequences.numpy():
        n = len(row)
        while n > 0 and row[n - 1] == pad_token_id:
            n -= 1
        seqs.append(row[:n].astype(np.int64))
    return seqs, [float(s) for s in out.sequences_scores]
