cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py > gpurun_out/r05/bench_a.json 2> gpurun_out/r05/bench_a.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r05/bench_a.json
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --no-cpu-baseline --no-north-star > gpurun_out/r05/bench_b.json 2>> gpurun_out/r05/bench_a.err; cut -c1-600 gpurun_out/r05/bench_b.json
