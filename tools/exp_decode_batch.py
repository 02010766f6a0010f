"""Developer experiment (round 6): beam-search decode time per image as a function of the number of images in ONE
search — the decode steps' GEMMs have M = 3 x images rows (10,752 at the bench's 3,584 images: one 128/256-row tile per CU, fill and
drain dominate), so a search over the images of several tower chunks should run nearer the MFMA / HBM rates.

    python tools/exp_decode_batch.py [images ...]      (default 3584 7168 10752)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd.blip import BLIP_Decoder  # noqa: E402
from vidil_amd.packing import set_compute_dtype  # noqa: E402
from vidil_amd.tokenizer import SyntheticBertTokenizer  # noqa: E402


def main():
    dev = torch.device("cuda")
    sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [3584, 7168, 10752]
    torch.manual_seed(0)
    cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=SyntheticBertTokenizer()).eval().to(dev)
    set_compute_dtype("bf16", cap)
    Te = 197
    for B in sizes:
        cap.__dict__.pop("_decode_state", None)
        torch.cuda.empty_cache()
        enc = (torch.randn(B * Te, 768, device=dev) * 0.5).to(torch.bfloat16)
        for _ in range(3):                           # eager, capture, replay
            cap.generate_ids(enc, B, num_beams=3, max_length=20, min_length=5)
        torch.cuda.synchronize()
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            cap.generate_ids(enc, B, num_beams=3, max_length=20, min_length=5)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"images {B:6d}  beam rows {3 * B:6d}  decode {dt * 1e3:8.1f} ms  = {dt / B * 1e6:7.2f} us / image   "
              f"(mem {torch.cuda.max_memory_allocated() / 2**30:.0f} GiB)", flush=True)
        del enc


if __name__ == "__main__":
    main()
