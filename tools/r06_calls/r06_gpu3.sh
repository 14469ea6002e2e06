cd $GRAFT_REPO_ROOT
echo "== gpu tests (expressions, algorithms, jit, capi)"
timeout 1500 python -m pytest tests/test_expressions.py tests/test_algorithms.py tests/test_jit.py tests/test_capi_symbols.py tests/test_slices_centroid_views.py -x -q -m gpu 2>&1 | tail -8
echo "== fused vs separate expression passes"
for f in 1 0; do PST_EXPR_FUSE=$f PST_EXPR_FUSE_REAL=$f timeout 300 python tools/exp_expr_fused.py 2>&1 | grep -v amdgpu.ids | tail -3; done
