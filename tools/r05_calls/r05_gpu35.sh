cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_expressions.py tests/test_jit.py -m gpu -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300; echo "rc=${PIPESTATUS[0]}"
