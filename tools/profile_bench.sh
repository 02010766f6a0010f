#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the SAME bench.py command; summaries are copied to profiles/ by hand.
# usage (on the GPU box, via gpurun): bash tools/profile_bench.sh <tag> [bench args...]
TAG=${1:-r1}; shift
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline --no-secondary}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$ARGS" > $OUT/cmdline.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py $ARGS --no-roofline > $OUT/bench_traced.json 2>/dev/null
python $R/tools/rocprof_summary.py $OUT/trace > $OUT/kernel_summary.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py $ARGS --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py $ARGS --no-roofline > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o s -- python $R/bench.py $ARGS --no-roofline > /dev/null 2>&1
python $R/tools/pmc_summary.py $OUT "$ARGS" > $OUT/pmc_summary.md
# gpurun merges at most 64 MiB back: keep the summaries and the per-kernel stats, drop the raw traces / counter dumps
find $OUT/trace -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/trace/t_kernel_stats.csv 2>/dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
ls $OUT; head -30 $OUT/kernel_summary.md; cat $OUT/bench.json | cut -c1-400
