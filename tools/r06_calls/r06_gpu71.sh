cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
# final tree: the deep fuzz module at 4x depth and 20 seeds of the sparse kNN fuzz
PST_DEEP_FUZZ=4 timeout 240 python -m pytest tests/test_deep_fuzz.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2 | tee gpurun_out/r06/deep_fuzz_final_tree.txt
timeout 100 python tools/fuzz_knn_sparse.py 20 1 2>&1 | tail -2 | tee -a gpurun_out/r06/deep_fuzz_final_tree.txt
