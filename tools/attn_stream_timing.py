"""Developer: where a wave of the streamed attention kernel spends its cycles (library built with -DVIDIL_ATTN_TIMING; the
kernel then overwrites the head of `out` with per-wave s_memtime deltas: total, at barrier B1, at barrier B2, in the store phase)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd import kernels as K  # noqa: E402

B, H, T = int(sys.argv[1]) if len(sys.argv) > 1 else 3584, 12, 197
dev = "cuda"
q = (torch.randn(B, H, T, 64, device=dev) * 0.125).bfloat16()
k = torch.randn(B, H, T, 64, device=dev).bfloat16()
v = torch.randn(B, H, T, 64, device=dev).bfloat16()
out = torch.empty(B * T, H * 64, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    K.attention(q, k, v, out, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K.attention(q, k, v, out, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=0)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
d = out.view(-1).view(torch.int64)[:512 * 8 * 4].view(512, 8, 4).cpu().double()
tot = d[:, :7, 0]
print(f"launch {us:.1f} us; wave total cycles mean {tot.mean():.0f} (=> {tot.mean() / us:.0f} MHz if a wave lives the whole launch)")
for i, name in ((1, "B1 wait"), (2, "B2 wait"), (3, "store phase")):
    x = d[:, :7, i]
    print(f"  {name:12s} mean {x.mean():9.0f} cycles = {100 * x.mean() / tot.mean():5.1f} % of a wave's life;  per wave 0..6: "
          + " ".join(f"{100 * x[:, w].mean() / tot[:, w].mean():.1f}" for w in range(7)))
