cd /root/repo
python -m pytest tests -x -q -m gpu -k "normals or knn or sparse" 2>&1 | tail -3
for w in normals_knn16_sheet normals_knn16; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'])"; done
python tools/fuzz_knn_sparse.py 40 11 2>&1 | tail -2
