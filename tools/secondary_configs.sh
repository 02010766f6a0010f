cd /tmp
run() { echo "\$ python bench.py $*"; python $GRAFT_REPO_ROOT/bench.py "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; c = d['config']
print(f\"{d['value']} frames/s {d['ms_per_step']} ms/step; dominant {r['kernel']} {r['achieved']} TFLOP/s of {r['peak']} frac {r['frac']} ; gflop/frame {c.get('algorithmic_gflop_per_frame', c.get('gflop_per_frame'))} whole-path frac {c.get('whole_path_mfma_frac')}\")
"; }
run --dtype f16 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run --dtype fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run --vit large --frames 16 --videos-per-step 96 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run --vit large --frames 16 --videos-per-step 96 --dtype fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run --size 384 --clip l14 --videos-per-step 64 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
