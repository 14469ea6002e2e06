cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_distributed_gloo.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "forced-rank|passed|failed" | tail -3; done
