"""Developer probe (not a test): HIP path vs the CPU oracle, prints error statistics."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from oracle import beam_ref, clip_ref, med_ref, vit_ref  # noqa: E402
from vidil_amd.blip import BLIP_Decoder, DecodeTrace  # noqa: E402
from vidil_amd.blip_itm import BLIP_ITM  # noqa: E402
from vidil_amd.clip import CLIPModel  # noqa: E402
from vidil_amd.tokenizer import SyntheticBertTokenizer  # noqa: E402


def perturb(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)


def stats(name, got, ref):
    d = (got - ref).abs()
    print(f"{name}: max|d|={d.max().item():.3e} mean|d|={d.mean().item():.3e} ref_absmax={ref.abs().max().item():.3f} "
          f"ref_std={ref.std().item():.3f}")


def main():
    dev = "cuda"
    torch.manual_seed(0)
    tok = SyntheticBertTokenizer()
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=tok)
    perturb(cap, 1)
    cap.eval()
    B = 4
    x = torch.randn(B, 3, 224, 224)
    sd = {k: v.clone() for k, v in cap.state_dict().items()}
    t0 = time.time()
    with torch.no_grad():
        y_ref = vit_ref.vit_forward(sd, x)
    print("oracle vit", time.time() - t0)
    capd = cap.to(dev)
    y32, y16 = capd.visual_encoder.forward_both(x.to(dev))
    torch.cuda.synchronize()
    stats("vit", y32.cpu(), y_ref)

    # decoder: prefill logits + full beam search
    trace = DecodeTrace()
    out_tok, out_len = capd.generate_ids(y16, B, num_beams=3, max_length=20, min_length=5, trace=trace)
    torch.cuda.synchronize()
    enc3 = y_ref.repeat_interleave(3, dim=0)
    state = {}
    otrace = []

    def step(ids, beam_idx):
        ids_t = torch.from_numpy(ids)
        with torch.no_grad():
            if beam_idx is None:
                lg, cache = med_ref.decoder_logits(sd, ids_t, enc3)
            else:
                past = med_ref.reorder_cache(state["cache"], torch.from_numpy(beam_idx))
                lg, cache = med_ref.decoder_logits(sd, ids_t, enc3, past)
        state["cache"] = cache
        return lg.numpy()

    prompt = capd.prompt_ids(B, "cpu").long().numpy()
    t0 = time.time()
    seqs, scores = beam_ref.beam_search(step, prompt, num_beams=3, max_length=20, min_length=5, eos_token_id=102,
                                        pad_token_id=0, trace=otrace)
    print("oracle beam", time.time() - t0)
    for s in range(len(trace.logits)):
        stats(f"logits step {s}", trace.logits[s].cpu(), torch.from_numpy(otrace[s]["logits"]))
        same = np.array_equal(trace.cand_index[s].cpu().numpy(), otrace[s]["cand_index"])
        print("   cand index equal:", same)
        if not same:
            break
    toks = out_tok.cpu().numpy()
    for b in range(B):
        ref = seqs[b]
        got = toks[b][: len(ref)]
        print("seq", b, "equal" if np.array_equal(got, ref) else f"DIFF\n  got {got}\n  ref {ref}")

    # ITM
    itm = BLIP_ITM(image_size=224, vit="base", tokenizer=tok)
    perturb(itm, 2)
    itm.eval()
    isd = {k: v.clone() for k, v in itm.state_dict().items()}
    with torch.no_grad():
        yi_ref = vit_ref.vit_forward(isd, x)
    caps = ["w2000 w2001 w2002", "w5 w6 w7 w8 w9 w10 w11 w12 w13", "a picture of w77", "w1234"]
    ids, lens = itm.tokenize(caps)
    am = (torch.arange(35)[None] < lens[:, None]).long()
    with torch.no_grad():
        ref_itm = med_ref.itm_logits(isd, yi_ref, ids.long(), am)
    itmd = itm.to(dev)
    got_itm = itmd(x.to(dev), caps)
    stats("itm", got_itm.cpu(), ref_itm)
    print(got_itm.cpu(), ref_itm)

    # CLIP
    clip = CLIPModel()
    perturb(clip, 3)
    clip.eval()
    csd = {k: v.clone() for k, v in clip.state_dict().items()}
    with torch.no_grad():
        ie_ref = clip_ref.image_embeds(csd, x)
    tids = torch.randint(1000, 40000, (6, 12))
    tids[:, 0] = 49406
    tids[:, -1] = 49407
    tids[2, 7:] = 49407
    with torch.no_grad():
        te_ref = clip_ref.text_embeds(csd, tids)
    clipd = clip.to(dev)
    ie = clipd.encode_image(x.to(dev))
    te = clipd.encode_text(tids.to(dev))
    stats("clip image", ie.cpu(), ie_ref)
    stats("clip text", te.cpu(), te_ref)


if __name__ == "__main__":
    main()
