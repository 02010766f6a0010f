"""Developer microbenchmark: GPU frame resize (Pillow-exact bicubic) for N frames of HxW -> 224^2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd.preprocess import blip_frames, clip_frames  # noqa: E402


def main():
    N, H, W = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 360, 640)))
    x = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda")
    for name, fn in (("blip squash", lambda: blip_frames(x, 224)), ("clip edge+crop", lambda: clip_frames(x, 224))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gb = N * (H * W * 3 + 224 * 224 * 3) / 1e9
        print(f"{name:15s} N={N} {H}x{W}: {ms:7.3f} ms  {N / ms * 1e3:9.0f} frames/s  {gb / ms * 1e3:7.1f} GB/s (in+out bytes)")


if __name__ == "__main__":
    main()
