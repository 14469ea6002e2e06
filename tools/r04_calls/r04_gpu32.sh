cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_capi_symbols.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head -8
