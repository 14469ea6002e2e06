cd $GRAFT_REPO_ROOT
for f in 0 0 0 0; do PST_EXPR_FUSE=$f timeout 300 python -X faulthandler tools/exp_expr_fused.py 20000000 > /tmp/o.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids /tmp/o.txt | tail -25 | cut -c1-250; done
