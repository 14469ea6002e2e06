cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python tools/exp_jit_cache_damage.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/jit_cache_damage.txt | cut -c1-330
