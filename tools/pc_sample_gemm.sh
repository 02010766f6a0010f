#!/bin/bash
# developer: PC-sample the big plain-f16 GEMM to see where gemm256 waves spend their time
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pcs; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/one_gemm.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from vidil_amd import kernels as K
M, N, Kd = 201728, 3072, 768
a = (torch.randn(M, Kd, device="cuda") * 0.5).half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half()
o = torch.empty(M, N, dtype=torch.float16, device="cuda")
for _ in range(30): K.gemm(a, w, None, out=o)
torch.cuda.synchronize()
PY
for method in stochastic host_trap; do
  unit=cycles; interval=1048576; [ $method = host_trap ] && unit=time && interval=100
  timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $interval --output-format csv -d $OUT/$method -o p -- python /tmp/one_gemm.py > $OUT/$method.log 2>&1
  echo "== $method rc=$?"; tail -3 $OUT/$method.log; find $OUT/$method -type f | head; 
done
