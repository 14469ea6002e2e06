cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 tools/bin/ts5 exp3 > gpurun_out/r05/ts5_exp3.txt 2>&1
echo "ts5 rc=$?"
grep -E "^occ2" gpurun_out/r05/ts5_exp3.txt | head -150
