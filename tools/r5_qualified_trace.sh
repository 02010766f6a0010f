#!/bin/bash
# developer: bench line + GEMM shape table + kernel trace of one precision configuration: tools/r5_qualified_trace.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --gemm-shapes > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_traced.json 2>/dev/null
python $R/tools/rocprof_summary.py $OUT/trace --top 45 > $OUT/kernel_summary.md; rm -rf $OUT/trace
cat $OUT/kernel_summary.md | cut -c1-160; cut -c1-300 $OUT/bench.json
