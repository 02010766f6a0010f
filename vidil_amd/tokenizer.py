"""Tokenizers at the text <-> id boundary.

The reference uses HuggingFace ``BertTokenizer('bert-base-uncased')`` plus two added
special tokens (models/blip.py:290-295).  ``init_tokenizer`` builds exactly that (from
``vocab_file`` / $VIDIL_BERT_VOCAB, a local HF cache, or the hub) and RAISES when it
cannot: a real checkpoint decoded through a stand-in vocabulary would write garbage
captions without any error.

``SyntheticBertTokenizer`` is an id-preserving stand-in for benchmarks and tests on
boxes without the vocabulary (token id N <-> the word ``wN``; the three prompt words
keep their real bert-base-uncased ids), so the whole string-level pipeline — decode,
prompt stripping, exact-match dedup, re-tokenisation for the ITM filter — still runs
and round-trips ids exactly.  It is used only on explicit request: pass
``tokenizer=SyntheticBertTokenizer()`` to the model constructors, or set
``VIDIL_TOKENIZER=synthetic``; the factories refuse it together with ``pretrained=``.
"""
from __future__ import annotations

import os

import torch

PAD, UNK, CLS, SEP, MASK = 0, 100, 101, 102, 103
BOS_DEC, ENC = 30522, 30523          # '[DEC]', '[ENC]' appended to the 30522-entry vocab
_KNOWN = {"a": 1037, "picture": 3861, "of": 1997, "video": 2678, "photo": 6302}
_KNOWN_INV = {v: k for k, v in _KNOWN.items()}
_SPECIAL = {PAD: "[PAD]", UNK: "[UNK]", CLS: "[CLS]", SEP: "[SEP]", MASK: "[MASK]", BOS_DEC: "[DEC]", ENC: "[ENC]"}
_SPECIAL_INV = {v: k for k, v in _SPECIAL.items()}


class Encoding(dict):
    """Minimal BatchEncoding: attribute access + ``.to(device)``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        return Encoding({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})


class SyntheticBertTokenizer:
    vocab_size = 30524
    pad_token_id, unk_token_id, cls_token_id, sep_token_id, mask_token_id = PAD, UNK, CLS, SEP, MASK
    bos_token_id = BOS_DEC
    enc_token_id = ENC
    additional_special_tokens_ids = [ENC]
    is_synthetic = True

    def __init__(self, allow_pretrained=False):
        # allow_pretrained: tests that exercise load_checkpoint on synthetic checkpoints; never for real ones
        self.allow_pretrained = allow_pretrained

    def add_special_tokens(self, mapping):
        return 0

    # -- text -> ids -------------------------------------------------------------------
    def _word_id(self, w):
        if w in _KNOWN:
            return _KNOWN[w]
        if w in _SPECIAL_INV:
            return _SPECIAL_INV[w]
        if len(w) > 1 and w[0] == "w" and w[1:].isdigit() and int(w[1:]) < self.vocab_size:
            return int(w[1:])
        return UNK

    def encode_one(self, text, max_length=None, truncation=False):
        ids = [CLS] + [self._word_id(w) for w in text.lower().split()] + [SEP]
        if truncation and max_length is not None and len(ids) > max_length:
            ids = ids[: max_length - 1] + [SEP]
        return ids

    def __call__(self, text, padding=False, truncation=False, max_length=None, return_tensors=None, **_):
        single = isinstance(text, str)
        seqs = [self.encode_one(t, max_length, truncation) for t in ([text] if single else text)]
        if padding in (True, "longest"):
            L = max(len(s) for s in seqs)
        elif padding == "max_length":
            L = max_length
        else:
            L = None
        masks = [[1] * len(s) for s in seqs]
        if L is not None:
            masks = [m + [0] * (L - len(m)) for m in masks]
            seqs = [s + [PAD] * (L - len(s)) for s in seqs]
        if return_tensors == "pt":
            return Encoding(input_ids=torch.tensor(seqs, dtype=torch.long), attention_mask=torch.tensor(masks, dtype=torch.long))
        if single:
            return Encoding(input_ids=seqs[0], attention_mask=masks[0])
        return Encoding(input_ids=seqs, attention_mask=masks)

    # -- ids -> text -------------------------------------------------------------------
    def decode(self, ids, skip_special_tokens=False):
        if torch.is_tensor(ids):
            ids = ids.tolist()
        out = []
        for i in ids:
            i = int(i)
            if i in _SPECIAL:
                if not skip_special_tokens:
                    out.append(_SPECIAL[i])
            elif i in _KNOWN_INV:
                out.append(_KNOWN_INV[i])
            else:
                out.append(f"w{i}")
        return " ".join(out)


def init_tokenizer(vocab_file=None):
    """Reference: models/blip.py:290-295 — BertTokenizer('bert-base-uncased') + '[DEC]' (bos) + '[ENC]'.

    Sources, in order: ``vocab_file`` / $VIDIL_BERT_VOCAB (a vocab.txt; a wrong path raises), the local HF cache,
    the hub (as the reference's ``from_pretrained`` does).  ``VIDIL_TOKENIZER=synthetic`` selects the stand-in
    explicitly.  Anything else that fails raises: there is no silent fallback."""
    if os.environ.get("VIDIL_TOKENIZER", "") == "synthetic":
        return SyntheticBertTokenizer()
    vocab_file = vocab_file or os.environ.get("VIDIL_BERT_VOCAB")
    from transformers import BertTokenizer

    if vocab_file:
        if not os.path.isfile(vocab_file):
            raise FileNotFoundError(f"init_tokenizer: vocabulary file {vocab_file!r} does not exist")
        with open(vocab_file, encoding="utf-8") as f:
            vocab = {w.rstrip("\n"): i for i, w in enumerate(f)}
        try:
            tok = BertTokenizer(vocab=vocab)               # transformers 5.x (vocab_file= is silently ignored there)
        except TypeError:
            tok = BertTokenizer(vocab_file=vocab_file)     # transformers 4.x, the reference's API
        if len(tok) != len(vocab):
            raise RuntimeError(f"init_tokenizer: BertTokenizer holds {len(tok)} entries, {vocab_file!r} has {len(vocab)}")
    else:
        try:
            tok = BertTokenizer.from_pretrained("bert-base-uncased", local_files_only=True)
        except Exception:
            try:
                tok = BertTokenizer.from_pretrained("bert-base-uncased")
            except Exception as e:
                raise RuntimeError(
                    "init_tokenizer: the bert-base-uncased vocabulary is not available (no local HF cache, no network). "
                    "Point $VIDIL_BERT_VOCAB at a vocab.txt, or — for benchmarks / tests with random weights only — "
                    "request the id-preserving stand-in explicitly: tokenizer=SyntheticBertTokenizer() or "
                    "VIDIL_TOKENIZER=synthetic.") from e
    tok.add_special_tokens({"bos_token": "[DEC]"})
    tok.add_special_tokens({"additional_special_tokens": ["[ENC]"]})
    tok.enc_token_id = tok.convert_tokens_to_ids("[ENC]")     # (== additional_special_tokens_ids[0] of the 4.x API)
    return tok


def refuse_synthetic_with_checkpoint(tokenizer, pretrained):
    """A real checkpoint with the stand-in vocabulary decodes to pseudo-words that the ITM filter re-tokenises
    consistently — garbage written to video_text_CapFilt.json without an error.  Refuse the combination."""
    if pretrained and getattr(tokenizer, "is_synthetic", False) and not getattr(tokenizer, "allow_pretrained", False):
        raise RuntimeError("a pretrained checkpoint needs the real bert-base-uncased tokenizer, not SyntheticBertTokenizer "
                           "(unset VIDIL_TOKENIZER=synthetic / pass a real tokenizer, or set $VIDIL_BERT_VOCAB)")
