"""developer experiment: what dropping finished images from the decode batch buys when captions end at different
lengths (they never do on the random-weight benchmark: a [SEP]-logit bias stands in for trained weights here).

    python tools/exp_decode_compaction.py [boosts...]          (GPU box)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_frames  # noqa: E402
from vidil_amd.blip import CLIP_MEAN, CLIP_STD  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    boosts = [float(x) for x in sys.argv[1:]] or [0.0, 1.3, 1.4, 1.45]
    cap, flt, clip, tok = build_models(dev, 224, "b32", "base", "bf16")
    cap = cap.to(dev)
    Nv, F = 384, 8
    frames = torch.from_numpy(synthetic_frames(Nv, F, 224, 0)).to(dev).reshape(Nv * F, 224, 224, 3)
    y16 = cap.visual_encoder.forward_u8(frames, CLIP_MEAN, CLIP_STD)[1]
    bias = cap.text_decoder.cls.predictions.bias
    sep = cap.tokenizer.sep_token_id
    for boost in boosts:
        with torch.no_grad():
            bias[sep] += boost
        res = {}
        for cm in (0, 256):
            cap.__dict__.pop("_decode_state", None)
            ts = []
            for _ in range(4):                       # eager, capture, 2 x replay
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                tk, ln = cap.generate_ids(y16, Nv * F, num_beams=3, max_length=20, min_length=5, compact_min=cm)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            used = sorted(k[0] for k in cap._decode_state if isinstance(k[-1], tuple))
            res[cm] = (min(ts[2:]), tk.cpu(), ln.cpu(), used)
        same = torch.equal(res[0][1], res[256][1]) and torch.equal(res[0][2], res[256][2])
        ln = res[0][2].float()
        print(f"[SEP] +{boost}: caption lengths mean {ln.mean():.1f} (min {int(ln.min())}, max {int(ln.max())}); decode whole batch "
              f"{res[0][0]:.1f} ms, with finished images leaving {res[256][0]:.1f} ms (sessions {res[256][3]}), identical tokens: {same}",
              flush=True)
        with torch.no_grad():
            bias[sep] -= boost


if __name__ == "__main__":
    main()
