"""Developer microbenchmark: BLIP ViT-B/16 forward over N frames, processed in chunks (MALL-residency experiment)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd.vit import VisionTransformer  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    chunks = [int(c) for c in sys.argv[2:]] or [1024, 512, 256, 128]
    torch.manual_seed(0)
    m = VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12).to("cuda")
    x = torch.randn(N, 197, 768, device="cuda")
    for c in chunks:
        def run():
            for i in range(0, N, c):
                m.run_blocks(x[i:i + c].reshape(-1, 768).clone(), min(c, N - i))
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"N={N} chunk={c:5d}: {ms:8.2f} ms  ({N / ms * 1e3:8.0f} frames/s, {33.5e9 * N / ms / 1e9:7.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
