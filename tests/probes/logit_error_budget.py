"""Developer probe: where the caption-logit error (vs the fp32 CPU oracle) comes from, per stage, for f16 and bf16 and
with / without the error-compensated LM head.  Prints the table recorded in DESIGN.md §4.

  full        : ViT -> cross K/V -> 12 decoder layers -> LM head, all on the device
  dec only    : the ORACLE's ViT output (rounded to T16) fed to the device decoder + head
  head only   : the ORACLE's last hidden states fed to the device LM head
"""
import os
import sys

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))

from common import perturb_, synthetic_frames  # noqa: E402
from oracle import clip_ref, med_ref, vit_ref  # noqa: E402
from vidil_amd.blip import BLIP_Decoder, DecoderSession  # noqa: E402
from vidil_amd.packing import set_compute_dtype  # noqa: E402
from vidil_amd.tokenizer import SyntheticBertTokenizer  # noqa: E402


def stats(d, ref):
    return f"max {d.abs().max().item():.2e}  mean {d.abs().mean().item():.2e}  (scale {ref.abs().max().item():.2f})"


def main():
    dev = "cuda"
    torch.manual_seed(0)
    tok = SyntheticBertTokenizer()
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=tok).eval()
    perturb_(cap, 100)
    sd = {k: v.clone() for k, v in cap.state_dict().items()}
    B, nb = 3, 3
    u8 = synthetic_frames(1, B)[0]
    x = clip_ref.preprocess_u8(u8)
    prompt = cap.prompt_ids(B, "cpu").long()
    with torch.no_grad():
        y_ref = vit_ref.vit_forward(sd, x)
        h_ref, _ = med_ref.bert_model(sd, "text_decoder.bert.", prompt, None, enc=y_ref, is_decoder=True)
        lg_ref = med_ref.lm_head(sd, "text_decoder.cls.", h_ref[:, -1])
    cap = cap.to(dev)
    for dt in ("f16", "bf16"):
        set_compute_dtype(dt, cap)
        tdt = torch.float16 if dt == "f16" else torch.bfloat16
        for precise in (False, True):
            cap.text_decoder.precise_head = precise
            y32, y16 = cap.visual_encoder.forward_u8(torch.from_numpy(u8).to(dev), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
            sess = DecoderSession(cap.text_decoder, y16, B, nb, 20)
            lg = sess.prefill(prompt.to(torch.int32).reshape(-1).to(dev), 4, shared=True).cpu()
            sess2 = DecoderSession(cap.text_decoder, y_ref.reshape(-1, y_ref.shape[-1]).to(dev).to(tdt), B, nb, 20)
            lg2 = sess2.prefill(prompt.to(torch.int32).reshape(-1).to(dev), 4, shared=True).cpu()
            h32 = h_ref.reshape(-1, h_ref.shape[-1]).to(dev).contiguous()
            lg3 = cap.text_decoder.lm_logits(h32.to(tdt), B, 4, h32=h32).cpu()
            tag = f"{dt} {'precise head' if precise else 'plain head  '}"
            print(f"{tag} | full      {stats(lg - lg_ref, lg_ref)}")
            print(f"{tag} | dec only  {stats(lg2 - lg_ref, lg_ref)}")
            print(f"{tag} | head only {stats(lg3 - lg_ref, lg_ref)}")
        print(f"{dt} ViT output: {stats(y32.cpu() - y_ref, y_ref)}")


if __name__ == "__main__":
    main()
