#!/bin/bash
# copies the summaries of gpurun_out/prof_<tag> (written by tools/profile_bench.sh on the GPU box) into profiles/
# usage: bash tools/copy_profiles.sh <tag> [name prefix, default r3]      e.g. copy_profiles.sh r3bf16 r3_bf16
P=gpurun_out/prof_$1
N=${2:-r4}
{ echo "# rocprofv3 --kernel-trace --stats of \`python bench.py $(head -1 $P/cmdline.txt 2>/dev/null) --no-roofline\` (1,792 videos x 8 frames per step in tower chunks of 896 unless the arguments say otherwise, 1 x MI355X)"; echo
  echo "Bench line of the traced run (tracing costs a few %): \`$(cut -c1-260 $P/bench_traced.json)...\`"; echo
  cat $P/kernel_summary.md; echo; echo "## rocprofv3 --stats (t_kernel_stats.csv, top 25)"; echo; echo '```'; head -26 $P/trace/t_kernel_stats.csv | cut -c1-200; echo '```'; } > profiles/${N}_bench_kernel_trace.md
{ echo "# rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_* in separate runs, each with --kernel-trace only) of the same bench command"; echo; cat $P/pmc_summary.md; } > profiles/${N}_bench_pmc.md
# (what bench.py reads `roofline.traffic` from: the latest profile) — stamped with the commit / date it was taken at
python3 - "$P/pmc_traffic.json" <<'PY'
import json, subprocess, sys, time
d = json.load(open(sys.argv[1]))
d["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
d["taken"] = time.strftime("%Y-%m-%d")
json.dump(d, open(sys.argv[1], "w"), indent=1)
PY
cp $P/pmc_traffic.json profiles/${N}_pmc_traffic.json
# profiles/pmc_traffic.json (what bench.py reads) holds the kernels of EVERY operand type profiled: a third argument "merge"
# adds this profile's kernels (e.g. the `gemm4w_kernel<fp8, ...>` names of a --dtype fp8 run) to the file instead of replacing it
if [ "$3" = "merge" ] && [ -f profiles/pmc_traffic.json ]; then
python3 - "$P/pmc_traffic.json" "$N" <<'PY'
import json, sys
new = json.load(open(sys.argv[1]))
cur = json.load(open("profiles/pmc_traffic.json"))
for k, v in new["kernels"].items():
    v["from_profile"] = sys.argv[2]
    cur["kernels"][k] = v
cur.setdefault("merged", []).append({"profile": sys.argv[2], "commit": new.get("commit"), "taken": new.get("taken"), "source": new.get("source")})
json.dump(cur, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
else
cp $P/pmc_traffic.json profiles/pmc_traffic.json
fi
cp $P/bench.json profiles/${N}_bench.json
