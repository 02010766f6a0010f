#!/bin/bash
# developer: where the wave cycles of the staged ViT self-attention kernel go (SQ counters, one pass; attention microbenchmark)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o sq -- python $GRAFT_REPO_ROOT/tools/bench_attn.py vit ${1:-1024} > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- python $GRAFT_REPO_ROOT/tools/bench_attn.py vit ${1:-1024} > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_attn'
for sub in ('sq','sq2'):
    agg=collections.defaultdict(list)
    for f in glob.glob(f'{out}/{sub}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_lds" not in r["Kernel_Name"] and "attn_stream" not in r["Kernel_Name"]: continue
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(sub, {c: round(sum(x)/len(x)) for c,x in agg.items()}, 'launches', len(next(iter(agg.values()), [])))
PY
