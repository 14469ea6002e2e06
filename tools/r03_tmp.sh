cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "normals or knn or sparse" 2>&1 | tail -4
for rep in 1 2; do for bl in 0 1; do
  PST_KNN_BOX_LIST=$bl python bench.py --workload normals_knn16_sheet --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SHEET boxlist=$bl', d['ms_per_step'])"
done; done
python bench.py --workload normals_knn16 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('UNIFORM default', d['ms_per_step'])"
PST_KNN_DEBUG=1 PST_KNN_TRACE=1 python bench.py --workload normals_knn16_sheet --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "pst knn" | tail -6
