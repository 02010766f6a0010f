#!/bin/bash
# quick kernel-trace summary of the bench (developer)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench.json 2>/dev/null
python $R/tools/rocprof_summary.py $OUT/trace --top 30 > $OUT/kernel_summary.md
cat $OUT/kernel_summary.md; cut -c1-200 $OUT/bench.json
