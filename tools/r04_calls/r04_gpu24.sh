cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "one_pass_fit" 2>&1 | grep -E "one-pass|passed|failed|Error" | tail -5
