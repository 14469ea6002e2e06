cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_jit.py -m gpu -x -q -k "tiny_and_largest" 2>&1 | tail -5
PST_FUZZ_SCALE=40 timeout 2000 python -m pytest tests/test_jit.py -m gpu -x -q -k "specialised_compaction_vs" 2>&1 | tail -5
