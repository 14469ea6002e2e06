cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_out_of_memory.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30 | cut -c1-400
# after other big-memory tests and the multi-thread tests in one process
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_out_of_memory.py -m gpu -q -p no:cacheprovider -k "release or 4gib or out_of_memory or full_device or failed_allocation or two_threads or 2_pow_32" 2>&1 | tail -8 | cut -c1-300
