cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_jit.py tests/test_filter_append.py -m gpu -x -q -k "tiny_and_largest or filter or compaction" 2>&1 | tail -5
