"""Developer microbenchmark (round 5): the parity mode's attention kernels at the bench's shapes.
  decode cross-attention, split-operand form on 16-bit K / V tiles (attn_direct_kernel QS; $VIDIL_ATTN_QS_VARIANT 0 / 1 / 2)
  against the plain 16-bit direct kernel and the f32 VALU kernel; the towers' self-attention: attn_split_kernel against the
  f32-MFMA kernel and the plain streamed kernel; the decode steps' self-attention over an f32 KV arena.
usage: python tools/bench_attn_split.py [images]"""
import os
import sys
import time

os.environ["VIDIL_DEV_ENV"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vidil_amd import kernels as K

dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
H, C, T = 12, 768, 197
nb = 3


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


g = torch.Generator(device="cpu").manual_seed(0)
# ---- decode cross-attention: nb beams x 1 token per image over 197 image keys
Tc = 224
kt = torch.randn(B, H, Tc, 64, generator=g).half().to(dev)
vt = torch.randn(B, H, Tc, 64, generator=g).half().to(dev)
q32 = torch.randn(B * nb, C, generator=g).to(dev)
q16 = (q32.view(B * nb, H, 1, 64) * 0.125).half().contiguous()
o3 = torch.zeros(B * nb, 3 * C, dtype=torch.float16, device=dev)
o16 = torch.zeros(B * nb, C, dtype=torch.float16, device=dev)
byts = 2 * B * H * T * 64 * 2
t = timeit(lambda: K.attention(q16, kt, vt, o16, Bq=B * nb, H=H, Nq=1, Nk=T, Tq_cap=1, Tk_cap=Tc, NP=Tc, kv_group=nb, kv_tiled=True))
print(f"decode cross-attention, {B} images: plain 16-bit direct kernel {t:8.1f} us  ({byts / t / 1e6:.2f} TB/s of K / V)")
for var in ("0", "2"):
    os.environ["VIDIL_ATTN_QS_VARIANT"] = var
    t = timeit(lambda: K.attention_f32(q32, kt, vt, o3, Bq=B * nb, H=H, Nq=1, Nk=T, kv_rows=Tc, kv_group=nb, arith=1, kv16=True))
    print(f"   split Q / P on the 16-bit tiles, variant {var}: {t:8.1f} us  ({byts / t / 1e6:.2f} TB/s)")
kv32 = torch.randn(B, T, 2 * C, generator=g).to(dev)
t = timeit(lambda: K.attention_f32(q32, kv32[..., :C], kv32[..., C:], o3, Bq=B * nb, H=H, Nq=1, Nk=T, kv_rows=T, kv_group=nb, arith=0), n=5, warm=2)
print(f"   f32 K / V rows, f32 VALU kernel: {t:8.1f} us;", end=" ")
t = timeit(lambda: K.attention_f32(q32, kv32[..., :C], kv32[..., C:], o3, Bq=B * nb, H=H, Nq=1, Nk=T, kv_rows=T, kv_group=nb, arith=1), n=5, warm=2)
print(f"split kernel on f32 rows: {t:8.1f} us  ({2 * byts / t / 1e6:.2f} TB/s)")
del kv32, kt, vt
# ---- tower self-attention
qkv = torch.randn(B * T, 3 * C, generator=g).to(dev)
o3t = torch.zeros(B * T, 3 * C, dtype=torch.float16, device=dev)
for nw in ("8", "4"):
    os.environ["VIDIL_ATTN_SPLIT_NW"] = nw
    t = timeit(lambda: K.attention_f32(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o3t, Bq=B, H=H, Nq=T, Nk=T, arith=1), n=5, warm=2)
    print(f"tower self-attention, {B} images: split-operand kernel, {nw} waves per workgroup {t:8.1f} us", end="; ")
del os.environ["VIDIL_ATTN_SPLIT_NW"]
t = timeit(lambda: K.attention_f32(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o3t, Bq=B, H=H, Nq=T, Nk=T, arith=0), n=3, warm=1)
print(f"f32-MFMA kernel {t:8.1f} us")
del qkv, o3t
# ---- decode self-attention over an f32 arena
R, Tcap = B * nb, 20
ak = torch.randn(Tcap, R, C, generator=g).to(dev)
av = torch.randn(Tcap, R, C, generator=g).to(dev)
anc = torch.randint(0, R, (R, Tcap), generator=g, dtype=torch.int32).to(dev)
for n_keys in (5, 12, 19):
    res = []
    for gather in ("1", "0"):
        os.environ["VIDIL_ATTN_F32_ARENA_GATHER"] = gather
        res.append(timeit(lambda: K.attention_f32(q32, ak, av, o3, Bq=R, H=H, Nq=1, Nk=n_keys, anc=anc, arena_rows=R)))
    print(f"arena self-attention, {R} rows, {n_keys} keys: gather form {res[0]:7.1f} us, one key at a time {res[1]:7.1f} us")
