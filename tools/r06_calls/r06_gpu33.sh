cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_algorithms.py tests/test_las_encode.py tests/test_buffer_converter.py tests/test_expressions.py tests/test_filter_append.py -m gpu -q -p no:cacheprovider -x -s 2>&1 | grep -E "expression fuzz|passed|failed|Error" | tail -6
for w in filter_las0_columnar filter_big_interleaved; do
  timeout 900 python tools/abab.py --workload $w --a "PST_RESULTS_TO_HOST=0" --b "PST_RESULTS_TO_HOST=1" --pairs 6 --steps 20 --out gpurun_out/r06/abab_results_to_host_$w.txt 2>&1 | tail -3
done
