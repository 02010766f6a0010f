#!/bin/bash
# developer: phase-staggered workgroups in gemm4w (a build with -DVIDIL_4W_STAGGER_EXP as build/ab/libvidil_stag.so)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out/stagger
export VIDIL_HIP_LIB=$R/build/ab/libvidil_stag.so
for cfg in "0 2" "40 2" "80 2" "160 2" "80 4" "160 4" "56 2" "0 2"; do
  set -- $cfg
  echo "== stagger period $1 half-us, $2 phases"
  VIDIL_4W_STAGGER=$1 VIDIL_4W_STAGGER_PH=$2 timeout 300 python tools/bench_gemm.py 7168 2>&1 | grep -E "qkv|proj|fc1|fc2|plain f16 "
done 2>&1 | tee gpurun_out/stagger/micro.txt
