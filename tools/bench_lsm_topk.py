"""Developer microbenchmark: log-softmax + top-2K of one decode step (10,752 beam rows x 30,524 logits, 3 beams)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidil_amd import kernels as K  # noqa: E402

B, nb, V = int(sys.argv[1]) if len(sys.argv) > 1 else 3584, 3, 30524
torch.manual_seed(0)
logits = torch.randn(B * nb, V, device="cuda")
bs = torch.randn(B * nb, device="cuda")
for _ in range(3):
    s, i = K.logsoftmax_topk(logits, bs, B, nb, 102)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    s, i = K.logsoftmax_topk(logits, bs, B, nb, 102)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"lsm_topk rows={B * nb} V={V}: {us:.1f} us  {B * nb * V * 4 / us / 1e6:.2f} TB/s  digest {float(s.double().sum()):.6f} {int(i.long().sum())}")
