"""Developer tool: wall time of each phase of one bench step (events on the stream), 128 videos x 8 frames."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vidil_amd.blip import CLIP_MEAN, CLIP_STD  # noqa: E402


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


def main():
    dev = torch.device("cuda")
    cap, flt, clip, tok = bench.build_models(dev)
    cap, flt, clip = cap.to(dev), flt.to(dev), clip.to(dev)
    Nv, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 384), 8
    frames = torch.from_numpy(bench.synthetic_frames(Nv, F, 224)).to(dev).reshape(Nv * F, 224, 224, 3)
    B = Nv * F
    t, (y32, y16) = timed(lambda: cap.visual_encoder.forward_u8(frames, CLIP_MEAN, CLIP_STD))
    print(f"BLIP ViT (one model)      {t:8.2f} ms")
    t, cross = timed(lambda: cap.text_decoder.bert.project_cross_kv(y16, B, 197))
    print(f"cross K/V projection      {t:8.2f} ms")
    t, out = timed(lambda: cap.generate_ids(y16, B, num_beams=3, max_length=20, min_length=5), n=2)
    print(f"caption decode (beam 3)   {t:8.2f} ms  (includes the cross K/V projection)")
    caps = cap.decode_captions(out[0])
    per_video = [list(dict.fromkeys(caps[v * F:(v + 1) * F])) for v in range(Nv)]
    from vidil_amd.capfilt import CapFiltEngine
    cfg = dict(caption=True, filter=True, filter_generated_only=True, keep_original_caption=False, threshold=0.4,
               filter_mode="max_filter", generation_mode="beam", do_sentence_tokenization=True, image_size=224, vit="base")
    eng = CapFiltEngine(cfg, dev, captioner=cap, filterer=flt)
    t, _ = timed(lambda: eng._filter_batch(frames, Nv, F, per_video), n=2)
    print(f"ITM filter (ViT + enc)    {t:8.2f} ms  ({sum(len(c) for c in per_video) * F} pairs)")
    t, _ = timed(lambda: clip.encode_image_u8(frames))
    print(f"CLIP ViT-B/32             {t:8.2f} ms")


if __name__ == "__main__":
    main()
