cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "into_async" 2>&1 | tail -25
