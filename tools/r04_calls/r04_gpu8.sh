cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_jit.py -m gpu -x -q -k "in_place" 2>&1 | tail -8
for v in 1 0; do echo "PST_TRANSFORM_WHOLE_RECORDS=$v"; PST_JIT=sync PST_TRANSFORM_WHOLE_RECORDS=$v python tools/exp_transform.py 2>&1 | tail -2; done
