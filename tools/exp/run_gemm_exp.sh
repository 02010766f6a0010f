#!/bin/bash
# developer experiment: rebuild gemm.o with EXP_* flags on the GPU box and time the decode shapes
cd $GRAFT_REPO_ROOT/vidil_amd/csrc
cp libvidil_hip.so /tmp/lib_keep.so
for e in "" "-DEXP_NOEPI" "-DEXP_NOLOOP"; do
  cp ../../tools/exp/gemm_exp.hip.txt gemm_exp.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVIDIL_GEMM_TUNE $e -c gemm_exp.hip -o gemm_exp.o 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvidil_hip.so core.o gemm_exp.o gemm256.o gemm256w4.o attention.o rowops.o beam.o scan.o
  echo "=== $e"
  (cd ../.. && timeout 200 python tools/tune_gemm.py 384 2>&1 | grep -v amdgpu)
done
cp /tmp/lib_keep.so libvidil_hip.so; rm -f gemm_exp.hip gemm_exp.o
