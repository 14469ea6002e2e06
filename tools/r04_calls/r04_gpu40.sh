cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "filter or padding or tiny or background" 2>&1 | tail -3
PST_FUZZ_SCALE=15 timeout 2000 python -m pytest tests/test_jit.py -m gpu -x -q -k "specialised_compaction_vs" 2>&1 | tail -3
