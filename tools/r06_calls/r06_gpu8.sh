cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_distributed_gloo.py tests/test_capi_symbols.py tests/test_algorithms.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -30 | cut -c1-400
