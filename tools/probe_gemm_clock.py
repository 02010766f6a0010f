"""Developer: the shader clock gemm256 actually runs at, per shape (needs `make EXTRA=-DVIDIL_GEMM_PROBE`).

Each shape loops for ~1.5 s (the power controller settles), then the in-kernel probe of the last launch is read:
sclk = s_memtime ticks / s_memrealtime ticks x 100 MHz.  "pipe busy" = achieved FLOP/s over what the matrix
pipes could do at THAT clock (256 CUs x 4 SIMDs x 1024 FLOP per cycle).
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd import _lib, kernels as K  # noqa: E402


def main():
    lib = _lib.load()
    probe = lib.vidil_debug_gemm_probe
    probe.restype = C.c_int
    probe.argtypes = [C.POINTER(C.c_ulonglong)]
    dev = "cuda"
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    M = B * 197
    shapes = [("proj/f32+res", M, 768, 768, "f32"), ("fc1/f16+gelu", M, 3072, 768, "gelu"), ("fc2/f32+res", M, 768, 3072, "f32"),
              ("plain f16", M, 3072, 768, "f16"), ("plain f16 K3072", M, 768, 3072, "f16")]
    for name, m, n, k, epi in shapes:
        a = (torch.randn(m, k, device=dev) * 0.5).half()
        w = (torch.randn(n, k, device=dev) * 0.05).half()
        bias = torch.randn(n, device=dev)
        if epi == "f32":
            x = torch.randn(m, n, device=dev)
            fn = lambda: K.gemm(a, w, bias, out=x, resid=x)  # noqa: E731
        elif epi == "gelu":
            o = torch.empty(m, n, dtype=torch.float16, device=dev)
            fn = lambda: K.gemm(a, w, bias, out=o, act=K.ACT_GELU_ERF)  # noqa: E731
        else:
            o = torch.empty(m, n, dtype=torch.float16, device=dev)
            fn = lambda: K.gemm(a, w, bias, out=o)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        calls = 0
        while time.time() - t0 < 1.5:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            calls += 20
        dt = (time.time() - t0) / calls
        buf = (C.c_ulonglong * 2)()
        assert probe(buf) == 0
        ghz = buf[0] / buf[1] * 0.1
        tf = 2.0 * m * n * k / dt / 1e12
        print(f"{name:16s} {dt * 1e6:8.1f} us  {tf:7.1f} TFLOP/s  sclk {ghz:.2f} GHz  pipe busy {100 * tf / (ghz * 1048.576):.0f}%")


if __name__ == "__main__":
    main()
