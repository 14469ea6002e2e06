cd $GRAFT_REPO_ROOT
( time PST_DEEP_FUZZ=15 timeout 3000 python -m pytest tests/test_deep_fuzz.py -m gpu -q 2>&1 | tail -15 ) 2>&1 | tail -20
