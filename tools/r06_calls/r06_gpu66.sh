cd $GRAFT_REPO_ROOT
PST_STRICT_OOM=1 timeout 900 python -m pytest tests/test_out_of_memory.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 | cut -c1-400
# with a parent that keeps 236 GB in its pool (the state at the end of the whole suite)
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_jit.py tests/test_out_of_memory.py -m gpu -q -p no:cacheprovider -k "2_pow_32_typed_las or out_of_memory or full_device or failed_allocation or release_scratch_hands or disk_cache" 2>&1 | tail -12 | cut -c1-400
