cd $GRAFT_REPO_ROOT
for f in 1 0 1 0; do PST_EXPR_FUSE=$f timeout 300 python -X faulthandler tools/exp_expr_fused.py 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-300; done
