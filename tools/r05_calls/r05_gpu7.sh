cd $GRAFT_REPO_ROOT
bash tools/exp_resident.sh
