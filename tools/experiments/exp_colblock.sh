#!/bin/bash
# developer experiment (round 5): a column-blocked tile walk in gemm4w (build with -DVIDIL_4W_COLBLOCK_EXP as
# vidil_amd/csrc/libvidil_hip_cb.so), $VIDIL_4W_COLBLOCK = column tiles per block; bench line + the fc1 / QKV rows of the shape table,
# then FETCH_SIZE of the fc1 instantiation for the row-major walk and one blocking.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; export VIDIL_DEV_ENV=1
OUT=$R/gpurun_out/colblock; mkdir -p $OUT
echo "== shipped library"; python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --gemm-shapes 2> $OUT/tree.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep -E "M= 706048 N=  3072 K=  768|M= 706048 N=  2304 K=  768" $OUT/tree.err
export VIDIL_HIP_LIB=$R/vidil_amd/csrc/libvidil_hip_cb.so
for CB in 0 2 3 4 6; do
  export VIDIL_4W_COLBLOCK=$CB
  echo "== experiment library, col block $CB"; python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --gemm-shapes 2> $OUT/cb$CB.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep -E "M= 706048 N=  3072 K=  768|M= 706048 N=  2304 K=  768" $OUT/cb$CB.err
done
for CB in 0 ${1:-4}; do
  export VIDIL_4W_COLBLOCK=$CB
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc$CB -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > /dev/null 2>&1
  python - $OUT/pmc$CB $CB <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm4w" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:110]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:4]:
    print(f"col block {sys.argv[2]}: FETCH_SIZE {sum(v) / len(v) / 1e6:8.2f} GB(raw KB/1e6; x2 = bytes) per launch over {len(v)} launches  {k}")
PY
  rm -rf $OUT/pmc$CB
done
