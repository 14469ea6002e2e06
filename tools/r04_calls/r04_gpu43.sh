cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do PST_FILTER_STREAM=$v python tools/exp_filter_padded.py narrow 2>&1 | grep -v amdgpu.ids; done; done
