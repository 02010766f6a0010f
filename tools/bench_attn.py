"""Developer microbenchmark: the cross-attention of one decode step (3 beams per image, 197 image tokens) — the
HBM-bound attn_direct kernel.  Prints time and the K + V^T bytes it streams per second."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd import kernels as K  # noqa: E402


def vit(B):
    """Encoder self-attention of a ViT-B/16 tower (197 tokens, 12 heads, V row-major): the staged kernel."""
    H, T = 12, 197
    dev = "cuda"
    torch.manual_seed(0)
    q = (torch.randn(B, H, T, 64, device=dev) * 0.125).half()
    k = torch.randn(B, H, T, 64, device=dev).half()
    v = torch.randn(B, H, T, 64, device=dev).half()
    o = torch.empty(B * T, H * 64, dtype=torch.float16, device=dev)
    fn = lambda: K.attention(q, k, v, o, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=0)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"vit self-attention images={B} {t * 1e6:8.1f} us  {4.0 * B * H * T * T * 64 / t / 1e12:.1f} TFLOP/s  {3 * B * H * T * 64 * 2 * 2 / t / 1e12:.2f} TB/s (Q, K, V in + O out)")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "vit":
        return vit(int(sys.argv[2]) if len(sys.argv) > 2 else 3584)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072        # images
    beams, H, Te = 3, 12, 197
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0            # key capacity of the buffers (0: tight)
    NP = cap or (Te + 15) // 16 * 16
    Tk = cap or Te
    dev = "cuda"
    torch.manual_seed(0)
    q = torch.randn(B * beams, H, 1, 64, device=dev).half()
    k = torch.randn(B, H, Tk, 64, device=dev).half()
    vt = torch.randn(B, H, 64, NP, device=dev).half()
    o = torch.empty(B * beams, H * 64, dtype=torch.float16, device=dev)
    fn = lambda: K.attention(q, k, vt, o, Bq=B * beams, H=H, Nq=1, Nk=Te, Tq_cap=1, Tk_cap=Tk, NP=NP, kv_group=beams)  # noqa: E731
    if len(sys.argv) > 3 and sys.argv[3] == "tiled":                 # fragment tiles (the product's decode layout)
        Tc = (Te + 31) // 32 * 32
        kt = torch.randn(B, H, Tc * 64, device=dev).half()
        vv = torch.randn(B, H, Tc * 64, device=dev).half()
        fn = lambda: K.attention(q, kt, vv, o, Bq=B * beams, H=H, Nq=1, Nk=Te, Tq_cap=1, Tk_cap=Tc, NP=0, kv_group=beams,  # noqa: E731
                                 kv_tiled=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 50
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    nbytes = B * H * (Te * 64 + 64 * Te) * 2
    print(f"images={B} {t * 1e6:8.1f} us  {nbytes / t / 1e12:.2f} TB/s (K + V^T, algorithmic)")


if __name__ == "__main__":
    main()
