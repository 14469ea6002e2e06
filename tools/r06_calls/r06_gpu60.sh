cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_out_of_memory.py tests/test_unaligned_external.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | cut -c1-400
