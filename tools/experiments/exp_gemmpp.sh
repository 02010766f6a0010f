#!/bin/bash
# Developer ablations of the ping-pong GEMM (csrc/gemmpp.hip): builds libvidil_hip_abl<N>.so beside the product library
# (VIDIL_PP_ABLATE: 1 = no LDS-DMA, 2 = no MFMA) and runs tools/bench_gemm.py against each through $VIDIL_HIP_LIB.
# Run the build part here (no GPU needed), the bench part on the GPU box:   tools/exp_gemmpp.sh build | bench
set -e
cd "$(dirname "$0")/../vidil_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
if [ "$1" = build ]; then
  for n in 1 2; do
    /opt/rocm/bin/hipcc $FLAGS -DVIDIL_PP_ABLATE=$n -c gemmpp.hip -o /tmp/gemmpp_abl$n.o
    objs=$(ls *.o | grep -v '^gemmpp.o$')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvidil_hip_abl$n.so $objs /tmp/gemmpp_abl$n.o
  done
else
  cd ../..
  echo "== product"; python tools/bench_gemm.py 512 | head -6
  for n in 1 2; do echo "== ablation $n"; VIDIL_HIP_LIB=$PWD/vidil_amd/csrc/libvidil_hip_abl$n.so python tools/bench_gemm.py 512 | head -6; done
  echo "== gemm256"; VIDIL_GEMMPP=0 python tools/bench_gemm.py 512 | head -6
fi
