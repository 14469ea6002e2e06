cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06/gpu_suite_final3.txt 2>&1
echo "suite exit $?" >> gpurun_out/r06/gpu_suite_final3.txt
grep -E "passed|failed|suite exit" gpurun_out/r06/gpu_suite_final3.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
