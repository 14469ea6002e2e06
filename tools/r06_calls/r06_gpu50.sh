cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_unaligned_external.py -m gpu -q -p no:cacheprovider 2>&1 | tail -60 | cut -c1-700 | tee gpurun_out/r06/unaligned_external.txt
