cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_many_attributes.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40 | cut -c1-600
