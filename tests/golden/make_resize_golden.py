"""Generates tests/golden/resize_golden.npz with the Pillow installed in the build container (run here, once):
inputs and the exact bytes ``PIL.Image.resize(..., BICUBIC)`` / HF-CLIP-style shortest-edge resize + centre crop
produce.  The reference applies these through torchvision / transformers (run_video_CapFilt.py:128-134,
run_visual_tokenization.py:138-142); Pillow's version is recorded in the file."""
import os

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def clip_style(img, S):
    h, w = img.shape[:2]
    nh, nw = (int(S * h / w), S) if w <= h else (S, int(S * w / h))
    r = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
    top, left = (nh - S) // 2, (nw - S) // 2
    return r[top:top + S, left:left + S]


def main():
    rng = np.random.default_rng(20260928)
    out = {"pillow_version": np.array(PIL.__version__)}
    cases = [("down", 90, 160, 64), ("up", 30, 41, 64), ("tall", 150, 70, 48), ("mixed", 100, 40, 64), ("same_h", 64, 100, 64)]
    for name, h, w, S in cases:
        noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = np.stack([xx * 255 // max(w - 1, 1), yy * 255 // max(h - 1, 1), (xx * 3 + yy * 5) % 256], -1).astype(np.uint8)
        for kind, img in (("noise", noise), ("smooth", smooth)):
            key = f"{name}_{kind}"
            out[key + "_in"] = img
            out[key + "_S"] = np.array(S)
            out[key + "_blip"] = np.asarray(Image.fromarray(img).resize((S, S), Image.BICUBIC))
            out[key + "_clip"] = clip_style(img, S)
    np.savez_compressed(os.path.join(HERE, "resize_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
