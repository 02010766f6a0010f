#!/bin/bash
# Same-box A/B of several builds of libvidil_hip.so (developer): tools/experiments/ab_libs.sh <tag> <lib> ...   ("tree" = the in-tree build)
# per library: the GEMM micro-benchmark at 3,584 frames, then the bench (3 steps) with its per-shape GEMM table; the first library once more at the end.
TAG=$1; shift
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
OUT=$R/gpurun_out/ab_$TAG; mkdir -p $OUT
for L in "$@" "$1"; do
  if [ "$L" = tree ]; then unset VIDIL_HIP_LIB; else export VIDIL_HIP_LIB=$R/$L; fi
  N=$(basename $L .so)
  echo "== $L micro"; timeout 300 python tools/bench_gemm.py 3584 2>&1 | head -6
  echo "== $L bench"; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --gemm-shapes 2> $OUT/$N.shapes.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])"
  grep -E "M= *(1412096|1021536|706048)" $OUT/$N.shapes.txt | head -12
done 2>&1 | tee $OUT/summary.txt
