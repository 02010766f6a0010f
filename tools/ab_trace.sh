#!/bin/bash
# A/B kernel traces of two trees on the same box (developer): tools/ab_trace.sh <tag> <treeA> <treeB>
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
for T in "$@"; do
  R=$GRAFT_REPO_ROOT/$T; N=$(echo $T | tr '/.' '__'); OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_$TAG/$N; mkdir -p $OUT
  (cd $R && python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200)
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench.json 2>/dev/null
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $OUT/trace --top 16 > $OUT/kernel_summary.md
  rm -rf $OUT/trace
  echo "== $T"; cat $OUT/kernel_summary.md; cut -c1-200 $OUT/bench.json
done
