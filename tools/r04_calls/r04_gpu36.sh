cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_jit.py -m gpu -x -q -k "background" 2>&1 | tail -3
PST_FUZZ_SCALE=20 timeout 1500 python -m pytest tests/test_jit.py -m gpu -x -q -k "specialised_compaction" 2>&1 | tail -3
PST_JIT=sync PST_FUZZ_SCALE=5 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_filter_append" 2>&1 | tail -3
