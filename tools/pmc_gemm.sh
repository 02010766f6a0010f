#!/bin/bash
# developer: PMC counters for the GEMM microbenchmark (separate passes, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
mkdir -p $OUT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/l2 -o l2 -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py 512 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py 512 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o sq -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py 512 > /dev/null 2>&1
find $OUT -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_gemm'
for sub in ('l2','fetch','sq'):
    files=glob.glob(f'{out}/{sub}/**/*counter_collection.csv', recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            if 'gemm256' not in k: continue
            agg[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(sub, k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'n=',len(next(iter(v.values()))))
PY
