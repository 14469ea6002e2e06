cd $GRAFT_REPO_ROOT
timeout 105 python -m pytest tests/test_deep_fuzz.py -m gpu -q -x --tb=short -k "knn" 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
