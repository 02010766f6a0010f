#!/bin/bash
# Same-box A/B of two builds of libvidil_hip.so inside the bench (developer): tools/ab_lib.sh <tag> <libA> <libB> ...
# (a path relative to the repo root, or "tree" for the in-tree build).  Boxes of the pool differ by +-3..5 %: only numbers
# taken in ONE gpurun call compare.  Untraced bench lines first (3 steps each, the first library once more at the end to
# show the drift), then the kernel-trace summary of each (2 steps).
TAG=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for L in "$@" "$1"; do
  if [ "$L" = tree ]; then unset VIDIL_HIP_LIB; else export VIDIL_HIP_LIB=$R/$L; fi
  echo "== $L  (untraced, 3 steps)"; python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline'))"
done
for L in "$@"; do
  if [ "$L" = tree ]; then unset VIDIL_HIP_LIB; else export VIDIL_HIP_LIB=$R/$L; fi
  N=$(basename $L .so); OUT=$R/gpurun_out/ab_$TAG/$N; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench.json 2>/dev/null
  python $R/tools/rocprof_summary.py $OUT/trace --top 14 > $OUT/kernel_summary.md
  rm -rf $OUT/trace
  echo "== $L (traced)"; cat $OUT/kernel_summary.md | cut -c1-150; cut -c1-200 $OUT/bench.json
done
