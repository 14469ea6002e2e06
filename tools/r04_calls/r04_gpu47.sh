cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_jit.py tests/test_filter_append.py -m gpu -x -q -k "filter or padding or tiny or compaction" 2>&1 | tail -3
PST_JIT=sync timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "las and (filter or round_trip)" 2>&1 | tail -3
