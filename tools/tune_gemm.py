"""Developer tool: time the decode-step GEMM shapes under every small-tile configuration.

Needs a library built with -DVIDIL_GEMM_TUNE (make EXTRA=-DVIDIL_GEMM_TUNE), which honours
VIDIL_GEMM_TILE=<BM>x<BN>x<ST>.  Usage: python tools/tune_gemm.py [rows ...]
"""
import os
os.environ.setdefault("VIDIL_DEV_ENV", "1")   # the library caches its developer switches per process otherwise
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd import kernels as K  # noqa: E402

CONFIGS = ["default", "128x128x2", "128x128x3", "128x128x4", "128x64x2", "128x64x3", "128x64x4", "64x64x2", "64x64x3",
           "64x64x4", "128x256x2", "128x256x3"]
if os.environ.get("VIDIL_TUNE_CONFIGS"):
    CONFIGS = os.environ["VIDIL_TUNE_CONFIGS"].split(",")


def timeit(fn, iters=30):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.1:       # (warm clocks: tools/bench_gemm.py has the story)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    rows = [int(x) for x in sys.argv[1:]] or [3072, 1536, 384]
    H = 12
    torch.manual_seed(0)
    for m in rows:
        shapes = [("qkv/heads T=1", 2304, 768, "heads3"), ("out f32+res", 768, 768, "f32"), ("cross-q heads", 768, 768, "heads1"),
                  ("fc1 gelu", 3072, 768, "gelu"), ("fc2 f32+res", 768, 3072, "f32")]
        for name, n, k, epi in shapes:
            a = (torch.randn(m, k, device=dev) * 0.5).half()
            w = (torch.randn(n, k, device=dev) * 0.05).half()
            bias = torch.randn(n, device=dev)
            if epi == "heads3":
                q = torch.empty(m, H, 1, 64, dtype=torch.float16, device=dev)
                kk = torch.empty(m, H, 32, 64, dtype=torch.float16, device=dev)
                vt = torch.empty(m, H, 64, 32, dtype=torch.float16, device=dev)
                hd = dict(q=q, k=kk, vt=vt, T=1, H=H, part0=0, t_off=7, Tq_cap=1, Tk_cap=32, NP=32, q_scale=0.125)
                fn = lambda: K.gemm(a, w, bias, heads=hd)  # noqa: E731
            elif epi == "heads1":
                q = torch.empty(m, H, 1, 64, dtype=torch.float16, device=dev)
                hd = dict(q=q, T=1, H=H, part0=0, t_off=0, Tq_cap=1, q_scale=0.125)
                fn = lambda: K.gemm(a, w, bias, heads=hd)  # noqa: E731
            elif epi == "f32":
                x = torch.randn(m, n, device=dev)
                fn = lambda: K.gemm(a, w, bias, out=x, resid=x)  # noqa: E731
            else:
                o = torch.empty(m, n, dtype=torch.float16, device=dev)
                fn = lambda: K.gemm(a, w, bias, out=o, act=K.ACT_GELU_ERF)  # noqa: E731
            line = f"M={m:5d} {name:14s} N={n:4d} K={k:4d} |"
            for cfg in CONFIGS:
                if cfg == "default":
                    os.environ.pop("VIDIL_GEMM_TILE", None)
                else:
                    os.environ["VIDIL_GEMM_TILE"] = cfg
                t = timeit(fn)
                line += f" {cfg}:{t * 1e6:6.1f}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
