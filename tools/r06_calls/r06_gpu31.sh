cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp ROUND=r06 COMMIT=2f73148
mkdir -p gpurun_out/r06
PST_FUZZ_SCALE=2 timeout 900 python -m pytest tests/test_expressions.py -m gpu -q -p no:cacheprovider -x -s -k "random_layouts_with_expression or ran_on_fused" 2>&1 | grep -E "expression fuzz|passed|failed|Error|assert" | tail -8
# the two kNN workloads again on the tree with the scratch-free reference-order fit (E11): kernel trace + FETCH / WRITE / SQ_INSTS_VALU passes
SPECS_FILE=tools/r06_specs_knn.txt bash tools/run_profiles.sh 2>&1 | grep -v "simple_timer" | tail -14
