cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "structured_volume or quantised_coordinates" 2>&1 | tail -12 | cut -c1-700
echo "=== guard off: what the tests say about the unguarded one-pass fit"
PST_KNN_FIT_GUARD=0 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "structured_volume" 2>&1 | grep -E "passed|failed|AssertionError: structured" | cut -c1-420 | tail -30
