cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head -10
