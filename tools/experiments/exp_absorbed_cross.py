"""Developer experiment (round 5): the WEIGHT-ABSORBED form of the decode steps' cross-attention, priced with
measurements before any kernel is written.

Today (per decoder layer and decode step, 3,584 images x 3 beams): attn_direct_kernel streams every image's K | V fragment tiles
(2 x 197 x 768 x 2 B = 605 KB per image) once: 2.2 GB per launch.  The absorbed form scores against the ENCODER STATES, which are the
same for all 12 layers: s = (q_h W_k,h) . enc^T, o_h = (P . enc) W_v,h^T (+ b_v; q . b_k is softmax-invariant) — 302 KB per image
instead of 605, and no cross K | V projection (5.58 GFLOP per frame, ~20 ms per step).  What it adds:
  * q~ = q_h W_k,h for every (beam row, head): 12 grouped GEMMs [rows, 64] x [64, 768] -> q~ [rows, 12, 768]   (write 198 MB)
  * the attention contracts over 768 instead of 64: 12 x the MFMA flops of today's kernel, and reads q~ (198 MB)
  * c = P . enc [rows, 12, 768] written (198 MB), then 12 grouped GEMMs [rows, 768] x [768, 64] read it (198 MB)
This script measures, on the box: today's kernel; the q~ / c traffic as plain device copies of the same size; the two grouped
GEMMs (torch.bmm: the library, tools only); an upper bound of the absorbed kernel's MFMA rate from a dense GEMM of its shape.
usage: python tools/experiments/exp_absorbed_cross.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vidil_amd import kernels as K

dev = "cuda"
B, nb, H, C, T = 3584, 3, 12, 768, 197
R = B * nb


def timeit(fn, n=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


g = torch.Generator(device="cpu").manual_seed(0)
Tc = 224
kt = torch.randn(B, H, Tc, 64, generator=g).bfloat16().to(dev)
vt = torch.randn(B, H, Tc, 64, generator=g).bfloat16().to(dev)
q16 = (torch.randn(R, H, 1, 64, generator=g) * 0.125).bfloat16().to(dev)
o16 = torch.zeros(R, C, dtype=torch.bfloat16, device=dev)
t_now = timeit(lambda: K.attention(q16, kt, vt, o16, Bq=R, H=H, Nq=1, Nk=T, Tq_cap=1, Tk_cap=Tc, NP=Tc, kv_group=nb, kv_tiled=True))
print(f"today: attn_direct_kernel on K | V tiles           {t_now:7.1f} us per launch  (2.20 GB -> {2.2e9 / t_now / 1e6:.2f} TB/s)")
del kt, vt
# the byte side of the absorbed form: enc read once per launch (1.08 GB) + q~ read + c written (198 MB each)
enc = torch.randn(B, T, C, generator=g).bfloat16().to(dev)
qt = torch.empty(R, H * C, dtype=torch.bfloat16, device=dev)
ct = torch.empty(R, H * C, dtype=torch.bfloat16, device=dev)
t_enc = timeit(lambda: enc.sum(dtype=torch.float32))                       # a read of enc at streaming rate (reduction: no write)
t_cpy = timeit(lambda: ct.copy_(qt))                                       # 198 MB read + 198 MB written
print(f"absorbed, bytes only: enc read {t_enc:7.1f} us (1.08 GB) + q~ read / c written {t_cpy:7.1f} us (2 x 198 MB)")
# the two grouped GEMMs around it (library, per head: [R, 64] x [64, 768] and [R, 768] x [768, 64])
qh = torch.randn(H, R, 64, generator=g).bfloat16().to(dev)
wk = torch.randn(H, 64, C, generator=g).bfloat16().to(dev)
ch = torch.randn(H, R, C, generator=g).bfloat16().to(dev)
wv = torch.randn(H, C, 64, generator=g).bfloat16().to(dev)
t_g1 = timeit(lambda: torch.bmm(qh, wk))
t_g2 = timeit(lambda: torch.bmm(ch, wv))
print(f"absorbed, grouped GEMMs: q~ = q W_k  {t_g1:7.1f} us,  o = c W_v^T {t_g2:7.1f} us   (12 heads, {R} rows; today's cross-query and "
      f"output projections stay as they are)")
# MFMA side: per image [36 rows -> 64] x [768] x [197 -> 208 keys], twice: as ONE batched GEMM of that shape (an upper bound: no softmax)
a = torch.randn(B, 64, C, generator=g).bfloat16().to(dev)
t_s = timeit(lambda: torch.bmm(a, enc.transpose(1, 2)))                     # scores  [64, 768] x [768, 197]
p_ = torch.randn(B, 64, T, generator=g).bfloat16().to(dev)
t_c = timeit(lambda: torch.bmm(p_, enc))                                    # context [64, 197] x [197, 768]
print(f"absorbed, contraction over 768 as batched library GEMMs (64-row tiles of 36 real rows): scores {t_s:7.1f} us, context {t_c:7.1f} us")
saved_proj = 20000.0 / (17 * 12)          # the cross K | V projection (~20 ms per step) spread over the 204 launches it serves
tot = max(t_enc + t_cpy, t_s + t_c) + t_g1 + t_g2
print(f"absorbed total >= max(bytes, MFMA) + grouped GEMMs = {tot:7.1f} us per launch against today's {t_now:7.1f} + {saved_proj:.0f} (projection share) "
      f"= {t_now + saved_proj:7.1f} us")
