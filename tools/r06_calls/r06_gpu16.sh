cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 3300 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r06/gpu_suite.txt
