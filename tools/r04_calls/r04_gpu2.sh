cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_jit.py -m gpu -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_jit.py 2>&1 | tail -5
GENERIC_ONLY=1 RANDOM_SEEDS="1" bash tools/r04_lines.sh b 2>&1 | tail -40
