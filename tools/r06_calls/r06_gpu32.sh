cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_algorithms.py tests/test_las_encode.py tests/test_buffer_converter.py tests/test_expressions.py -m gpu -q -p no:cacheprovider -x -s 2>&1 | grep -E "expression fuzz|passed|failed|Error" | tail -6
for w in bounds las0_encode; do
  timeout 900 python tools/abab.py --workload $w --a "PST_RESULTS_TO_HOST=0" --b "PST_RESULTS_TO_HOST=1" --pairs 6 --steps 20 --out gpurun_out/r06/abab_results_to_host_$w.txt 2>&1 | tail -5
done
