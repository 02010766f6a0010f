#!/bin/bash
# developer (round 6): socket power / shader clock while ONE fp8 GEMM shape of the tower mode loops — is the e4m3
# kernel (0.30-0.38 of the 5 PFLOP/s fp8 peak in the bench) at the socket's power cap like the 16-bit shapes (profiles/r2_power_probe.txt)?
# usage (GPU box): bash tools/power_probe_fp8.sh   -> one block per shape
cat > /tmp/loop_shape_fp8.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from vidil_amd import kernels as K
from vidil_amd.packing import FP8, w8
name = sys.argv[1]
M = 706048                                   # one tower chunk of the bench: 3,584 frames x 197 tokens
shapes = dict(fc1=(3072, 768, "f8gelu"), proj=(768, 768, "f32"), fc2=(768, 3072, "f32"), bf16_fc2=(768, 3072, "bf16"))
N, Kd, epi = shapes[name]
dev = "cuda"
if epi == "bf16":
    a = (torch.randn(M, Kd, device=dev) * 0.5).bfloat16(); w = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); x = torch.randn(M, N, device=dev)
    fn = lambda: K.gemm(a, w, bias, out=x, resid=x)
    peak = 2500.0
else:
    a = (torch.randn(M, Kd, device=dev) * 0.5).clamp(-448, 448).to(FP8)
    w, s = w8(torch.randn(N, Kd) * 0.05)
    w, s = w.to(dev), s.to(dev)
    bias = torch.randn(N, device=dev)
    peak = 5000.0
    if epi == "f32":
        x = torch.randn(M, N, device=dev); fn = lambda: K.gemm(a, w, bias, out=x, resid=x, w_scale=s, dtype16=torch.bfloat16)
    else:
        o = torch.empty(M, N, dtype=FP8, device=dev); fn = lambda: K.gemm(a, w, bias, out=o, act=K.ACT_GELU_ERF, w_scale=s, dtype16=torch.bfloat16)
print(name, K.gemm_kernel_name(a, w, bias, **({"out": x, "resid": x} if epi != "f8gelu" else {"out": o, "act": K.ACT_GELU_ERF}),
                               **({} if epi == "bf16" else {"w_scale": s, "dtype16": torch.bfloat16})), flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < 6:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
tf = 2.0 * M * N * Kd * n / dt / 1e12
print(f"{name}: {n} GEMMs in {dt:.2f}s: {tf:.1f} TFLOP/s sustained = {tf / peak:.3f} of the {peak:.0f} TFLOP/s peak, {dt / n * 1e6:.0f} us each")
PY
for s in fc1 fc2 proj bf16_fc2; do
  python /tmp/loop_shape_fp8.py $s &
  PID=$!
  sleep 3.6
  for i in 1 2; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4; sleep 1.0; done
  wait $PID
  echo "--"
done
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4
