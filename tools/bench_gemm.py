"""Developer microbenchmark: the GEMM shapes of the hot path, per epilogue (TFLOP/s, random data)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd import kernels as K  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    T, H = 197, 12
    M = B * T
    torch.manual_seed(0)
    shapes = [("qkv/heads", M, 2304, 768, "heads"), ("proj/f32+res", M, 768, 768, "f32"), ("fc1/f16+gelu", M, 3072, 768, "gelu"),
              ("fc2/f32+res", M, 768, 3072, "f32"), ("plain f16", M, 768, 768, "f16"), ("plain f16 big", M, 3072, 768, "f16"),
              ("lm_head f32", 1536, 30524, 768, "f32n"), ("decode 64x64", 1536, 768, 768, "f32"),
              ("decode ffn", 1536, 3072, 768, "gelu")]
    for name, m, n, k, epi in shapes:
        a = (torch.randn(m, k, device=dev) * 0.5).half()
        w = (torch.randn(n, k, device=dev) * 0.05).half()
        bias = torch.randn(n, device=dev)
        if epi == "heads":
            b_ = m // T
            q = torch.empty(b_, H, T, 64, dtype=torch.float16, device=dev)
            kk = torch.empty_like(q)
            vt = torch.empty(b_, H, 64, 208, dtype=torch.float16, device=dev)
            hd = dict(q=q, k=kk, vt=vt, T=T, H=H, part0=0, Tq_cap=T, Tk_cap=T, NP=208, q_scale=0.125)
            fn = lambda: K.gemm(a, w, bias, heads=hd)  # noqa: E731
        elif epi == "f32":
            x = torch.randn(m, n, device=dev)
            fn = lambda: K.gemm(a, w, bias, out=x, resid=x)  # noqa: E731
        elif epi == "f32n":
            x = torch.empty(m, n, device=dev)
            fn = lambda: K.gemm(a, w, bias, out=x)  # noqa: E731
        elif epi == "gelu":
            o = torch.empty(m, n, dtype=torch.float16, device=dev)
            fn = lambda: K.gemm(a, w, bias, out=o, act=K.ACT_GELU_ERF)  # noqa: E731
        else:
            o = torch.empty(m, n, dtype=torch.float16, device=dev)
            fn = lambda: K.gemm(a, w, bias, out=o)  # noqa: E731
        t = timeit(fn)
        print(f"{name:16s} M={m:6d} N={n:5d} K={k:4d}  {t * 1e6:8.1f} us  {2.0 * m * n * k / t / 1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
