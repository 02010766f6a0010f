#!/bin/bash
# developer experiment: rebuild gemm256 with EXP variants on the GPU box and time them
cd $GRAFT_REPO_ROOT/vidil_amd/csrc
for e in 0 6 7 8; do
  sed "s#/root/repo/vidil_amd/csrc/common.h#common.h#" ../../tools/gemm256_exp.hip.txt > gemm256_exp.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DEXP=$e -c gemm256_exp.hip -o gemm256.o 2>&1 | grep -E "error" 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvidil_hip.so core.o gemm.o gemm256.o gemm256w4.o attention.o rowops.o beam.o scan.o
  echo "=== EXP=$e"
  (cd ../.. && timeout 100 python tools/bench_gemm.py 512 2>&1 | grep -E "plain f16|fc2|lm_head")
done
rm -f gemm256_exp.hip
