"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32 / numpy / C) of the reference algorithm on
VidIL's frame-encoding hot path.  Every function cites the reference file:line
it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker / the
timed CPU baseline — never as the product path.  ``vidil_amd`` does not import
it and has no CPU fallback.

Pinning status:
  * vit_ref / med_ref  — pinned against the reference's own ``models/vit.py`` and
    ``models/med.py`` imported through ``oracle/ref_shim.py`` in the build
    container (tests/test_oracle_vs_reference.py, tests/golden/*.npz).
  * clip_ref           — pinned against ``transformers`` 5.15 ``CLIPModel`` (the
    reference calls HF CLIP, which is not vendored in the reference tree).
  * beam_ref           — PARITY UNPINNED against executable reference code: the
    beam search lives in ``transformers`` 4.15 ``generation_utils`` which is not
    installable here (the installed 5.15 ranks hypotheses differently).  It is a
    restatement of the published 4.15 algorithm, pinned only by known-answer
    tests on hand-built logit tables.
  * tokens_ref / scan_ref.c — restate run_visual_tokenization.py; the ontology
    filter is replayed on the reference's own JSON files (sizes recorded in
    tests/golden/ontology_sizes.json).
"""
