cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
# final tree of the round: the whole GPU suite, smoke(), the default bench line
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15; echo "suite exit ${PIPESTATUS[0]}" ) | tee gpurun_out/r06/gpu_suite_final8.txt | tail -6


