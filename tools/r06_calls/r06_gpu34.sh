cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
# the deep differential fuzz at four times the suite's default depth (seeds the suite never runs), final tree
PST_DEEP_FUZZ=12 timeout 3300 python -m pytest tests/test_deep_fuzz.py -m gpu -q -p no:cacheprovider > gpurun_out/r06/deep_fuzz_x12.txt 2>&1
echo "exit $?" >> gpurun_out/r06/deep_fuzz_x12.txt
grep -E "passed|failed|exit" gpurun_out/r06/deep_fuzz_x12.txt | tail -3
