cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_jit.py -m gpu -x -q -k "padding or tiny" 2>&1 | tail -5
PST_FUZZ_SCALE=20 timeout 2000 python -m pytest tests/test_jit.py -m gpu -x -q -k "specialised_compaction_vs" 2>&1 | tail -5
PST_JIT=sync PST_FUZZ_SCALE=3 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_filter_append.py -m gpu -x -q -k "filter" 2>&1 | tail -3
