"""Generates tests/golden/config1_*.json — BASELINE config 1, LITERALLY, as the files the reference's two scripts write
(run_video_CapFilt.py:261-291 -> video_text_CapFilt.json / video_text_Cap.json; run_visual_tokenization.py:447-463 ->
visual_tokens.json): 16 synthetic videos x 8 frames 224^2 (default_rng(1000 + video)), BLIP ViT-B/16 caption (beam 3) + CapFilt
+ CLIP ViT-B/32 visual tokens against an ontology with the vg category sizes (19,958 / 15,026 / 365 / 7,410), computed by the
fp32 CPU ORACLE (oracle/pipeline_ref.py: the reference's per-video loops) in this build container.

Weights (no checkpoint can be downloaded here): tests/common.portable_init_ — every parameter drawn from numpy's PCG64
stream, which is bit-identical on every host (torch's seeded CPU normal sampler is NOT: it differs between AVX2 and AVX-512
machines, i.e. between this container and the GPU box).  The golden records a checksum of every state dict; the GPU test
rebuilds the weights and refuses to compare if the checksums differ.  Ontology text embeddings: unit vectors from the same
kind of stream (the CLIP tokenizer's vocabulary is a download too), with one duplicated scene row (314 of the real file's 365
scene strings are distinct).

Beside the three documents it writes config1_margins.json: for every decision of the oracle, how close it was —
  caption_gap[video][frame]    min gap between adjacent beam candidates over the search (a near-tie may flip on the device)
  filter_margin[video][i]      |max_f p(caption i, frame f) - threshold|
  token_gap[video][frame][cat] min gap between adjacent scores among the top 6 (ranks closer than the fp32 summation-order
                               resolution are not decided by the reference form either)
so that the test can be EXACT outside a bounded, counted set of flagged entries.

Run:  python tests/golden/make_config1_golden.py          (~10-15 min on 8 cores)"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import portable_init_, synthetic_frames  # noqa: E402
from oracle import clip_ref, pipeline_ref, tokens_ref  # noqa: E402

VG = dict(objects=19958, attributes=15026, scenes=365, verbs=7410)
N_VIDEOS, F, THRESHOLD = 16, 8, 0.4


def ontology(dim=512, seed=3, sizes=VG):
    rng = np.random.default_rng(seed)
    emb, texts = {}, {}
    for k, n in sizes.items():
        e = torch.from_numpy(rng.standard_normal((n, dim), dtype=np.float32))
        emb[k] = e / e.norm(dim=-1, keepdim=True)
        texts[k] = [f"{k}{i}" for i in range(n)]
    emb["scenes"][10] = emb["scenes"][3]        # duplicate class strings -> exact score ties
    texts["scenes"][10] = texts["scenes"][3]
    return emb, texts


def state_checksum(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        if sd[k].dtype == torch.float32:
            h.update(k.encode())
            h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


def build_models():
    from vidil_amd.blip import BLIP_Decoder
    from vidil_amd.blip_itm import BLIP_ITM
    from vidil_amd.clip import CLIPModel
    from vidil_amd.tokenizer import SyntheticBertTokenizer

    tok = SyntheticBertTokenizer()
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=tok).eval()
    itm = BLIP_ITM(image_size=224, vit="base", tokenizer=tok).eval()
    clip = CLIPModel().eval()
    for i, m in enumerate((cap, itm, clip)):
        portable_init_(m, 100 + i)
    return tok, cap, itm, clip


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    tok, cap, itm, clip = build_models()
    sds = [{k: v.clone() for k, v in m.state_dict().items()} for m in (cap, itm, clip)]
    emb, texts = ontology()
    u8 = synthetic_frames(N_VIDEOS, F)
    prompt = cap.prompt_ids(1, "cpu")[0].long().numpy()
    filt, unfilt, toks = {}, {}, {}
    margins = dict(caption_gap={}, filter_margin={}, token_gap={})
    t0 = time.time()
    for v in range(N_VIDEOS):
        vid = f"video{v}"
        x = clip_ref.preprocess_u8(u8[v])
        otrace = []
        caps_frames = pipeline_ref.caption_video(sds[0], x, prompt, tok, cap.prompt, trace=otrace, dedup=True)
        gaps = np.stack([np.min(t["cand_scores"][:, :-1] - t["cand_scores"][:, 1:], axis=1) for t in otrace])     # [steps, F]
        margins["caption_gap"][vid] = [float(g) for g in gaps.min(axis=0)]
        caps = tokens_ref.dedup_captions(caps_frames)
        kept, probs = pipeline_ref.filter_video(sds[1], x, caps, tok, THRESHOLD, return_probs=True, dedup=True)
        margins["filter_margin"][vid] = [float(abs(float(np.max(p)) - THRESHOLD)) for p in probs]
        unfilt[vid] = caps
        if kept:                                  # run_video_CapFilt.py:190-204: a video with no caption left is not written
            filt[vid] = kept
        with torch.no_grad():
            ie = clip_ref.image_embeds(sds[2], x)
        scores = {k: (ie @ emb[k].t()).numpy() for k in tokens_ref.CATEGORIES}
        toks[vid] = tokens_ref.visual_tokens_from_scores([vid], [caps], scores, texts, F, 5)[vid]
        tg = []
        for f in range(F):
            row = {}
            for k in tokens_ref.CATEGORIES:
                top = np.sort(scores[k][f])[::-1][:6]
                row[k] = [float(top[r] - top[r + 1]) for r in range(5)]        # gap below rank r
            tg.append(row)
        margins["token_gap"][vid] = tg
        print(f"{vid}: {len(caps)} captions, {len(kept)} kept, {time.time() - t0:.0f}s", flush=True)
    meta = dict(videos=N_VIDEOS, frames=F, threshold=THRESHOLD, ontology_sizes=VG, torch=torch.__version__,
                checksums=dict(cap=state_checksum(sds[0]), itm=state_checksum(sds[1]), clip=state_checksum(sds[2])),
                generated_by="tests/golden/make_config1_golden.py (oracle/pipeline_ref.py, fp32 CPU)")
    margins["meta"] = meta
    for name, obj in (("config1_capfilt.json", filt), ("config1_cap.json", unfilt), ("config1_visual_tokens.json", toks),
                      ("config1_margins.json", margins)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=1)
    print("written", meta)


if __name__ == "__main__":
    main()
