cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
# final tree of the round: the whole GPU suite, smoke(), the default bench line
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15; echo "suite exit ${PIPESTATUS[0]}" ) | tee gpurun_out/r06/gpu_suite_final7.txt | tail -6
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r06/bench_line_final_f.json 2> gpurun_out/r06/bench_line_final_f.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r06/bench_line_final_f.json
