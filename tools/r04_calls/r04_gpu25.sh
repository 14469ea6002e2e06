cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q -k "rehearsed or forced" 2>&1 | tail -25
