cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_out_of_memory.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30 | cut -c1-400
# the same file after other big-memory tests in one process (the pool is not empty then), and the tests that use release_scratch
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_out_of_memory.py -m gpu -q -p no:cacheprovider -k "release or 4gib or out_of_memory or full_device or failed_allocation" 2>&1 | tail -40 | cut -c1-300
