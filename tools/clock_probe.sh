#!/bin/bash
# developer: sample GPU clock / power while the big f16 GEMM loops (is the MFMA rate power-limited?)
R=$GRAFT_REPO_ROOT
cat > /tmp/loop_gemm.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from vidil_amd import kernels as K
M, N, Kd = 201728, 3072, 768
a = (torch.randn(M, Kd, device="cuda") * 0.5).half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half()
o = torch.empty(M, N, dtype=torch.float16, device="cuda")
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    for _ in range(50): K.gemm(a, w, None, out=o)
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print(f"{n} GEMMs in {dt:.2f}s: {2.0*M*N*Kd*n/dt/1e12:.1f} TFLOP/s sustained")
PY
python /tmp/loop_gemm.py &
PID=$!
sleep 3
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8; echo --; sleep 1.2; done
wait $PID
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4
