cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_knn_guard.py 2>&1 | tail -40
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "structured_volume or quantised_coordinates" 2>&1 | tail -12 | cut -c1-500
