cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
PST_FUZZ_SCALE=8 timeout 2400 python -m pytest tests/test_expressions.py -m gpu -q -p no:cacheprovider -x -k "random_layouts" > gpurun_out/r06/expr_fuzz.txt 2>&1
echo "exit $?" >> gpurun_out/r06/expr_fuzz.txt
tail -30 gpurun_out/r06/expr_fuzz.txt
