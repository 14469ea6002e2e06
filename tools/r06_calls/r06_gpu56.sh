cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_out_of_memory.py -m gpu -q -s -p no:cacheprovider -k "release or 4gib or out_of_memory or full_device or failed_allocation" 2>&1 | grep -v amdgpu.ids | grep -a "DIAG\|passed\|failed" | cut -c1-300
