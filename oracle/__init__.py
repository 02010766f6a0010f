"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32 / numpy / C) of the reference algorithm on
VidIL's frame-encoding hot path.  Every function cites the reference file:line
it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker / the
timed CPU baseline — never as the product path.  ``vidil_amd`` does not import
it and has no CPU fallback.

Pinning status (DESIGN.md §4 has the details):
  * vit_ref / med_ref  — pinned against the reference's own ``models/vit.py`` and
    ``models/med.py`` imported through ``oracle/ref_shim.py`` in the build
    container (tests/test_oracle_cpu.py::test_oracle_vs_reference_full_size_modules,
    golden vectors tests/golden/*.npz written by tests/golden/make_golden.py).
  * clip_ref           — pinned against ``transformers`` 5.15 ``CLIPModel`` (the
    reference calls HF CLIP, which is not vendored in the reference tree).
  * beam_ref           — the search mechanics (log-softmax, EOS ban, 2 x beams
    candidates, beam selection, hypothesis banking, output format) are pinned
    against the EXECUTABLE installed ``transformers`` (5.15) through
    ``oracle/hf_beam.py`` / tests/test_beam_hf.py with ``rule="5.15"``; the three
    places where 4.15 (the version models/med.py names, not installable here)
    differs — hypothesis-score normaliser, the "no improvement possible" test,
    how the last step is banked — are restated from the published source and
    pinned by the known-answer tables of tests/beam_cases.py.
  * tokens_ref / scan_ref.c — restate run_visual_tokenization.py; the ontology
    filter is replayed on the reference's own JSON files (sizes recorded in
    tests/golden/ontology_sizes.json); scan_ref.c is tied to the reference FORM
    (``embeds @ text.T`` + ``argsort``) by tests/test_scan_ref_cpu.py.
  * sample_ref         — this library's own Philox draw contract (torch.multinomial's
    stream cannot be reproduced): PARITY UNPINNED for the draw, pinned for the
    logits processors / warpers by hand-derived cases.
"""
