#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out/pshape
for cfg in "448 448" "896 448" "448 448"; do
  set -- $cfg
  echo "== parity (all three models): $1 videos per step, tower chunks of $2"
  timeout 600 python bench.py --precision parity --videos-per-step $1 --tower-chunk-videos $2 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline 2> gpurun_out/pshape/err_$1_$2.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('peak_device_memory_gib'))" || tail -3 gpurun_out/pshape/err_$1_$2.txt
done 2>&1 | tee gpurun_out/pshape/summary.txt
( time python bench.py ) > gpurun_out/pshape/bench.json 2> gpurun_out/pshape/bench.err; grep -E "qualified|timed region" gpurun_out/pshape/bench.err; tail -3 gpurun_out/pshape/bench.err
