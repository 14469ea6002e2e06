cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06/gpu_suite_final2.txt 2>&1
echo "suite exit $?" >> gpurun_out/r06/gpu_suite_final2.txt
grep -E "passed|failed|suite exit" gpurun_out/r06/gpu_suite_final2.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py > gpurun_out/r06/bench_line_final.json 2> gpurun_out/r06/bench_line_final.err
echo "bench exit $?"; tail -c 1500 gpurun_out/r06/bench_line_final.json
