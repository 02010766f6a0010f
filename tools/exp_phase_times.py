"""developer experiment: does a tower run slower inside the step (between decode / ITM phases) than back to back?
HIP-event times of the filter ViT (a) looped by itself, (b) alternating with the caption decode, (c) after an idle gap.

    python tools/exp_phase_times.py [dtype]            (GPU box)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_frames  # noqa: E402
from vidil_amd.blip import CLIP_MEAN, CLIP_STD  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    cap, flt, clip, tok = build_models(dev, 224, "b32", "base", dt)
    cap, flt = cap.to(dev), flt.to(dev)
    Nv, F = 384, 8
    frames = torch.from_numpy(synthetic_frames(Nv, F, 224, 0)).to(dev).reshape(Nv * F, 224, 224, 3)
    y16 = cap.visual_encoder.forward_u8(frames, CLIP_MEAN, CLIP_STD)[1]

    def vit():
        return flt.visual_encoder.forward_u8(frames, CLIP_MEAN, CLIP_STD)[1]

    def decode():
        return cap.generate_ids(y16, Nv * F, num_beams=3, max_length=20, min_length=5)[0]

    def ev_time(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        return a, b

    for _ in range(3):
        decode(); vit()
    torch.cuda.synchronize()
    ts = [ev_time(vit) for _ in range(6)]
    torch.cuda.synchronize()
    print(f"{dt} ViT back to back      :", " ".join(f"{a.elapsed_time(b):6.1f}" for a, b in ts), flush=True)
    tv, td = [], []
    for _ in range(6):
        td.append(ev_time(decode)); tv.append(ev_time(vit))
    torch.cuda.synchronize()
    print(f"{dt} ViT after a decode    :", " ".join(f"{a.elapsed_time(b):6.1f}" for a, b in tv), flush=True)
    print(f"{dt} decode after a ViT    :", " ".join(f"{a.elapsed_time(b):6.1f}" for a, b in td), flush=True)
    ts = [ev_time(decode) for _ in range(4)]
    torch.cuda.synchronize()
    print(f"{dt} decode back to back   :", " ".join(f"{a.elapsed_time(b):6.1f}" for a, b in ts), flush=True)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); time.sleep(0.3)
        ts.append(ev_time(vit))
    torch.cuda.synchronize()
    print(f"{dt} ViT after 0.3 s idle  :", " ".join(f"{a.elapsed_time(b):6.1f}" for a, b in ts), flush=True)


if __name__ == "__main__":
    main()
