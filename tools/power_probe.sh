#!/bin/bash
# developer: socket power / shader clock while ONE GEMM shape of the path loops (is that kernel at the power cap?)
# usage (GPU box): bash tools/power_probe.sh   -> one block per shape
R=$GRAFT_REPO_ROOT
cat > /tmp/loop_shape.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from vidil_amd import kernels as K
name = sys.argv[1]
M = 201728
shapes = dict(plain=(3072, 768, "f16"), proj=(768, 768, "f32"), fc1=(3072, 768, "gelu"), fc2=(768, 3072, "f32"))
N, Kd, epi = shapes[name]
a = (torch.randn(M, Kd, device="cuda") * 0.5).half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half()
bias = torch.randn(N, device="cuda")
if epi == "f32":
    x = torch.randn(M, N, device="cuda"); fn = lambda: K.gemm(a, w, bias, out=x, resid=x)
elif epi == "gelu":
    o = torch.empty(M, N, dtype=torch.float16, device="cuda"); fn = lambda: K.gemm(a, w, bias, out=o, act=K.ACT_GELU_ERF)
else:
    o = torch.empty(M, N, dtype=torch.float16, device="cuda"); fn = lambda: K.gemm(a, w, None, out=o)
t0 = time.time(); n = 0
while time.time() - t0 < 6:
    for _ in range(50): fn()
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print(f"{name}: {n} GEMMs in {dt:.2f}s: {2.0*M*N*Kd*n/dt/1e12:.1f} TFLOP/s sustained, {dt/n*1e6:.0f} us each")
PY
for s in plain proj fc1 fc2; do
  python /tmp/loop_shape.py $s &
  PID=$!
  sleep 3.2
  for i in 1 2; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4; sleep 1.0; done
  wait $PID
  echo "--"
done
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4
