cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_algorithms.py -m gpu -x -q -k "normal or knn or sparse or into_async" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
timeout 900 python tools/fuzz_knn_sparse.py 60 9000 2>&1 | tail -2
for w in normals_knn16_sheet normals_knn16; do
  python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', 'ms_per_step', d['ms_per_step'])"
done
