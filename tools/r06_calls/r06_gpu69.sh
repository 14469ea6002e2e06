cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
# the full-device test's STRICT half, alone on a box (every refusal status 22, the 2.4-GB request refused), and the probe once more on the final tree
( PST_STRICT_OOM=1 timeout 900 python -m pytest tests/test_out_of_memory.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
  timeout 600 python tools/exp_oom.py 2>&1 | grep -v "amdgpu.ids" ) | tee gpurun_out/r06/oom_strict.txt | cut -c1-250
