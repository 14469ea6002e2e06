cd $GRAFT_REPO_ROOT
echo "== gpu tests (expressions, filter)"
timeout 1500 python -m pytest tests/test_expressions.py tests/test_filter_append.py -x -q -m gpu 2>&1 | tail -6
for f in 1 0 1 0; do PST_EXPR_FUSE=$f timeout 300 python tools/exp_filter_expr.py 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-300; done
