cd $GRAFT_REPO_ROOT
( time PST_DEEP_FUZZ=15 timeout 3000 python -m pytest tests/test_deep_fuzz.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" ) 2>&1 | tail -6
timeout 1500 python tools/fuzz_knn_sparse.py 200 7000 2>&1 | tail -4
