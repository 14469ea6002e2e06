cd $GRAFT_REPO_ROOT
bash tools/r06_lines.sh ${BOXTAG:-boxA} 2>&1 | tail -70
