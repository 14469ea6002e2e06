cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do
  timeout 900 python -m pytest tests/test_distributed_gloo.py tests/test_filter_append.py -m gpu -x -q 2>&1 | tail -25 | grep -v "^\.\|^$" | head -40
done
