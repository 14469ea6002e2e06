cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "structured_volume or quantised_coordinates" 2>&1 | tail -25
echo "=== the same with the guard off (expected: the forced one-pass fit leaves the window on the structures)"
PST_KNN_FIT_GUARD=0 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "structured_volume" 2>&1 | grep -E "passed|failed|AssertionError" | cut -c1-400 | tail -20
