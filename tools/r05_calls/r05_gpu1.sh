cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 tools/bin/ts5 all > gpurun_out/r05/ts5_all.txt 2>&1
echo "ts5 rc=$?"
grep -E "^check|^shape" gpurun_out/r05/ts5_all.txt | head -80
