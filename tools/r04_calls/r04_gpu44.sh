cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_jit.py tests/test_filter_append.py tests/test_deep_fuzz.py -m gpu -x -q -k "filter or padding or tiny or background or compaction" 2>&1 | tail -3
