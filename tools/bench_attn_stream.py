"""Developer A/B: the towers' self-attention on the streamed kernel against the staged one (same process, same operands).
    VIDIL_DEV_ENV=1 python tools/bench_attn_stream.py [images ...]
Prints µs per launch (HIP events on the launch stream, 20 launches after 3 warm-ups) and the traffic rate of Q + K + V + O."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VIDIL_DEV_ENV", "1")
from vidil_amd import kernels as K  # noqa: E402


def run(B, H=12, T=197, dtype=torch.bfloat16, iters=20):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    q = (torch.randn(B, H, T, 64, device=dev, generator=g) * 0.125).to(dtype)
    k = torch.randn(B, H, T, 64, device=dev, generator=g).to(dtype)
    v = torch.randn(B, H, T, 64, device=dev, generator=g).to(dtype)
    out = torch.empty(B * T, H * 64, dtype=dtype, device=dev)
    res = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["VIDIL_ATTN_STREAM"] = mode
        for _ in range(3):
            K.attention(q, k, v, out, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            K.attention(q, k, v, out, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=0)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        res.setdefault(mode, []).append(us)
    byts = 4 * B * H * T * 64 * 2
    flop = 4 * B * H * T * T * 64
    for mode, name in (("0", "staged"), ("1", "streamed")):
        us = min(res[mode])
        print(f"B={B} H={H} {name:9s} {us:8.1f} us  {byts / us / 1e6:5.2f} TB/s  {flop / us / 1e6:6.0f} TFLOP/s   (runs: "
              + ", ".join(f"{x:.1f}" for x in res[mode]) + ")")


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 3584]
    for b in sizes:
        run(b)
    run(1536, H=16)
