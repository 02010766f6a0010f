#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the SAME bench.py command; summaries are copied to profiles/ by hand.
# usage (on the GPU box, via gpurun): bash tools/profile_bench.sh <tag> [bench args...]
TAG=${1:-r1}; shift
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline}
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
python - <<PY > $OUT/pmc_summary.md
import csv, glob, collections, re
out="$OUT"
def short(n):
    n=re.sub(r"\(anonymous namespace\)::","",n); n=re.sub(r"^void ","",n); return re.sub(r"\(.*$","",n)[:70]
rows=collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch","pmc_write","pmc_sq"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
print("| kernel | launches | FETCH_SIZE KB/launch (x2 = bytes read, gfx950 correction) | WRITE_SIZE KB/launch | MFMA busy / (SQ_BUSY*32 SIMD/SE) | LDS conflict / active |")
print("|---|---:|---:|---:|---:|---:|")
for k,v in sorted(rows.items(), key=lambda kv:-sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES",[0]))):
    n=len(v.get("FETCH_SIZE",[])) or 1
    f=sum(v.get("FETCH_SIZE",[0]))/n; w=sum(v.get("WRITE_SIZE",[0]))/max(1,len(v.get("WRITE_SIZE",[])))
    mf=sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES",[0])); sb=sum(v.get("SQ_BUSY_CYCLES",[0]))
    lc=sum(v.get("SQ_LDS_BANK_CONFLICT",[0])); la=sum(v.get("SQ_LDS_IDX_ACTIVE",[0]))
    util = mf/(sb*32) if sb else 0
    print(f"| {k} | {n} | {f:.0f} | {w:.0f} | {util:.3f} | {lc/la if la else 0:.3f} |")
    traffic[k] = {"launches": n, "fetch_kb_raw": round(f), "write_kb": round(w),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024), "mfma_util": round(util, 3)}
import json
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of 'bench.py " + "$ARGS" + "'; FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); written by tools/profile_bench.sh",
           "kernels": traffic}, open(out + "/pmc_traffic.json", "w"), indent=1)
PY
ls $OUT; head -30 $OUT/kernel_summary.md; cat $OUT/bench.json | cut -c1-400
