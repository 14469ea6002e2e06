cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python tools/exp_oom.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r06/oom_probe.txt | cut -c1-260
echo "rc=${PIPESTATUS[0]}"
