cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 tools/bin/ts5 exp2 > gpurun_out/r05/ts5_exp2.txt 2>&1
echo "ts5 rc=$?"
grep -E "^check|^streams|^occ|^modes" gpurun_out/r05/ts5_exp2.txt | head -150
