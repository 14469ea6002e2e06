cd $GRAFT_REPO_ROOT
timeout 75 python tools/fuzz_knn_sparse.py 24 501 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
