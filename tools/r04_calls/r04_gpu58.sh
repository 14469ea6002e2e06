cd $GRAFT_REPO_ROOT
PST_JIT=0 timeout 900 python -m pytest tests -m gpu -x -q -k "filter and not streaming_kernels and not in_tree and not specialised and not compaction" 2>&1 | tail -3
PST_FILTER_SCAN_BLOCKS=0 PST_FILTER_STREAM=0 timeout 900 python -m pytest tests/test_filter_append.py tests/test_gpu_parity.py -m gpu -x -q -k "filter and not streaming_kernels and not in_tree" 2>&1 | tail -3
