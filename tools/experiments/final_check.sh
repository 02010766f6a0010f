R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out/final
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/final/gpu_tests.txt 2>&1
tail -5 gpurun_out/final/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/final/smoke.txt
( time python bench.py ) > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -c 1500 gpurun_out/final/bench.json; tail -5 gpurun_out/final/bench.err
