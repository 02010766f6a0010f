"""Developer probe (not a test; VERDICT r3 #2c): the cheapest parity MIX that keeps "caption logits within 1e-3" ABSOLUTE.

For k = 0, 2, 4, 6, 8, 10, 12 of the ViT's 12 blocks on error-compensated operands (the decoder trunk, the cross K|V
projection and the LM head always are): max|logit - fp32 oracle| over the 16 teacher-forced passes of a beam search
(3 frames), and the time of the caption path (ViT + beam-3 decode) on `--frames` frames.  Writes
gpurun_out/parity_mix.json.  Usage: python tests/probes/probe_parity_mix.py [--frames 256]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import perturb_, synthetic_frames  # noqa: E402
from oracle import beam_ref, clip_ref, med_ref, vit_ref  # noqa: E402
from vidil_amd.blip import BLIP_Decoder, DecoderSession  # noqa: E402
from vidil_amd.packing import set_compute_dtype, set_parity_mode  # noqa: E402
from vidil_amd.tokenizer import SyntheticBertTokenizer  # noqa: E402

DEV = "cuda"


def time_it(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    nfr = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 256
    torch.manual_seed(0)
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=SyntheticBertTokenizer()).eval()
    perturb_(cap, 100)
    sd = {k: v.clone() for k, v in cap.state_dict().items()}
    cap = cap.to(DEV)
    set_compute_dtype("f16", cap)
    B, nb = 3, 3
    u8 = synthetic_frames(1, B)[0]
    with torch.no_grad():
        y_ref = vit_ref.vit_forward(sd, clip_ref.preprocess_u8(u8))
    enc3 = y_ref.repeat_interleave(nb, dim=0)
    state, otrace, calls = {}, [], []

    def step(ids, beam_idx):
        calls.append((ids.copy(), None if beam_idx is None else beam_idx.copy()))
        with torch.no_grad():
            past = None if beam_idx is None else med_ref.reorder_cache(state["cache"], torch.from_numpy(beam_idx))
            lg, state["cache"] = med_ref.decoder_logits(sd, torch.from_numpy(ids), enc3, past)
        return lg.numpy()

    prompt = cap.prompt_ids(B, "cpu").long().numpy()
    beam_ref.beam_search(step, prompt, num_beams=nb, max_length=20, min_length=5, eos_token_id=102, pad_token_id=0, trace=otrace)
    big = torch.from_numpy(synthetic_frames(nfr // 8, 8, first_video=50).reshape(nfr, 224, 224, 3)).to(DEV)

    def errors():
        _, yop = cap.visual_encoder.forward_u8(torch.from_numpy(u8).to(DEV), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
        sess = DecoderSession(cap.text_decoder, yop, B, nb, 20)
        worst = []
        for s, (ids, beam_idx) in enumerate(calls[:len(otrace)]):
            if s == 0:
                lg = sess.prefill(torch.from_numpy(ids).to(torch.int32).reshape(-1).to(DEV), ids.shape[1])
            else:
                lg = sess.step(torch.from_numpy(ids[:, -1].copy()).to(torch.int32).to(DEV),
                               torch.from_numpy(beam_idx).to(torch.int32).to(DEV), ids.shape[1] - 1)
            worst.append((lg.cpu() - torch.from_numpy(otrace[s]["logits"])).abs().max().item())
        return worst

    def caption_path():
        _, yop = cap.visual_encoder.forward_u8(big, clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
        return cap.generate_ids(yop, nfr, num_beams=nb, max_length=20, min_length=5)

    rows = []
    e = errors()
    t = time_it(caption_path)
    rows.append(dict(mode="plain f16", vit_blocks_compensated=0, worst=max(e), per_pass=e, caption_path_ms=t * 1e3))
    print(rows[-1]["mode"], f"worst {max(e):.2e}  {t * 1e3:.1f} ms")
    t_plain = t
    set_parity_mode(True, cap)
    for k in (0, 2, 4, 6, 8, 10, 12):
        cap.visual_encoder.set_parity_last_blocks(None if k == 12 else k)
        cap.__dict__.pop("_decode_state", None)
        e = errors()
        t = time_it(caption_path)
        rows.append(dict(mode="parity: decoder + head + cross K|V, ViT last k blocks", vit_blocks_compensated=k, worst=max(e), per_pass=e,
                         caption_path_ms=t * 1e3, slowdown_vs_plain=t / t_plain))
        print(f"k={k:2d}: worst {max(e):.2e} (passes {min(e):.1e}..{max(e):.1e})  {t * 1e3:.1f} ms  x{t / t_plain:.2f}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_mix.json"), "w") as f:
        json.dump(dict(frames=nfr, logit_scale=float(np.abs(otrace[0]["logits"]).max()), rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
