#!/bin/bash
# copies the summaries of gpurun_out/prof_<tag> (written by tools/profile_bench.sh on the GPU box) into profiles/
P=gpurun_out/prof_$1
{ echo "# rocprofv3 --kernel-trace --stats of \`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline\` (defaults: 384 videos x 8 frames per step, 1 x MI355X)"; echo
  echo "Bench line of the traced run (tracing costs a few %): \`$(cut -c1-260 $P/bench_traced.json)...\`"; echo
  cat $P/kernel_summary.md; echo; echo "## rocprofv3 --stats (t_kernel_stats.csv, top 25)"; echo; echo '```'; head -26 $P/trace/t_kernel_stats.csv | cut -c1-200; echo '```'; } > profiles/r1_bench_kernel_trace.md
{ echo "# rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_* in separate runs, each with --kernel-trace only) of the same bench command"; echo; cat $P/pmc_summary.md; } > profiles/r1_bench_pmc.md
cp $P/pmc_traffic.json profiles/pmc_traffic.json
cp $P/bench.json profiles/r1_bench.json
