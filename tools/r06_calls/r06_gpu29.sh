cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06/gpu_suite_final.txt 2>&1
echo "suite exit $?" >> gpurun_out/r06/gpu_suite_final.txt
tail -5 gpurun_out/r06/gpu_suite_final.txt
timeout 900 python tools/fuzz_knn_sparse.py 20 77 > gpurun_out/r06/fuzz_sparse_final.txt 2>&1
echo "fuzz exit $?" >> gpurun_out/r06/fuzz_sparse_final.txt
tail -3 gpurun_out/r06/fuzz_sparse_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
