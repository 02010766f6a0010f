"""Per-tile overhead vs per-K-tile cost: plain 16-bit GEMM at M = 100864, N = 3072 over a sweep of K (bf16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vidil_amd import kernels as K
from tools.bench_gemm import timeit
M, N = 100864, 3072
tiles_per_cu = (M // 256) * (N // 256) / 256
for k in (128, 256, 512, 768, 1536, 3072):
    a = (torch.randn(M, k, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    t = timeit(lambda: K.gemm(a, w, None, out=o))
    tl = timeit(lambda: torch.matmul(a, w.t(), out=o))
    print(f"K={k:5d} nk={k//64:3d}  ours {t*1e6:8.1f} us = {t*1e6/tiles_per_cu:6.2f} us/tile {2.0*M*N*k/t/1e12:7.1f} TF   lib {tl*1e6:8.1f} us = {tl*1e6/tiles_per_cu:6.2f} us/tile")
