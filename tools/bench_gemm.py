"""Developer microbenchmark: the GEMM shapes of the hot path, per epilogue (TFLOP/s, random data).

`--calibrate` additionally times the LIBRARY GEMM (hipBLASLt / rocBLAS behind torch.matmul — tools only, never on the
product path) on the same box, same shapes, same random operands, bare (no bias, no activation, no residual): what the
silicon and its power cap give a tuned library kernel, to price "what 1.4 kW buys" against a measurement."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd import kernels as K  # noqa: E402


def timeit(fn, iters=40, warm_s=0.25):
    """The shader clock takes ~60 launches (~40 ms) of a big GEMM to ramp from idle: the first 20-launch block of a cold GPU
    reads 775 TFLOP/s where the settled one reads 930 (tools/experiments/exp_fc2_bistable.py).  Warm up by TIME, not by count,
    so that two kernels timed one after the other are compared at the same clock."""
    import time
    t0 = time.time()
    while time.time() - t0 < warm_s:
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


def calibrate(dev, M, dtypes=(torch.float16, torch.bfloat16)):
    """torch.matmul (hipBLASLt) on the path's three tower shapes + ours, bare 16-bit output, side by side."""
    for dt in dtypes:
        for name, m, n, k in (("qkv", M, 2304, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072), ("proj", M, 768, 768)):
            a = (torch.randn(m, k, device=dev) * 0.5).to(dt)
            w = (torch.randn(n, k, device=dev) * 0.05).to(dt)
            wt = w.t()                                   # [K, N] view: A @ W^T, the NT form nn.Linear uses
            o = torch.empty(m, n, dtype=dt, device=dev)
            t_lib = timeit(lambda: torch.matmul(a, wt, out=o))
            t_own = timeit(lambda: K.gemm(a, w, None, out=o))
            f = 2.0 * m * n * k / 1e12
            print(f"calibrate {str(dt)[6:]:8s} {name:5s} M={m:6d} N={n:5d} K={k:4d}  hipBLASLt {t_lib * 1e6:8.1f} us {f / t_lib:7.1f} TFLOP/s"
                  f"   {K.gemm_kernel_name(a, w, None, out=o).split('_kernel')[0]:7s} {t_own * 1e6:8.1f} us {f / t_own:7.1f} TFLOP/s   ratio {t_lib / t_own:5.2f}")


def main():
    dev = "cuda"
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(args[0]) if args else 512
    if "--shape" in sys.argv:            # --shape M N K: plain 16-bit-out GEMM of that shape, ours vs the library
        i = sys.argv.index("--shape")
        m, n, k = (int(v) for v in sys.argv[i + 1:i + 4])
        torch.manual_seed(0)
        for dt in (torch.float16, torch.bfloat16):
            a = (torch.randn(m, k, device=dev) * 0.5).to(dt)
            w = (torch.randn(n, k, device=dev) * 0.05).to(dt)
            o = torch.empty(m, n, dtype=dt, device=dev)
            t_lib = timeit(lambda: torch.matmul(a, w.t(), out=o))
            t_own = timeit(lambda: K.gemm(a, w, None, out=o))
            f = 2.0 * m * n * k / 1e12
            print(f"shape {m}x{n}x{k} {str(dt)[6:]:8s} hipBLASLt {f / t_lib:7.1f} TFLOP/s   {K.gemm_kernel_name(a, w, None, out=o).split('<')[0]} {f / t_own:7.1f} TFLOP/s")
        return
    if "--calibrate" in sys.argv:
        torch.manual_seed(0)
        calibrate(dev, B * 197)
        if "--only-calibrate" in sys.argv:
            return
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
