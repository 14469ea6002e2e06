cd $GRAFT_REPO_ROOT
B="--steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-extra-legs"
for i in 1 2 3; do python bench.py --workload normals_knn16_sheet $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sheet', l['ms_per_step'])"; done
PST_KNN_FIT=seq python bench.py --workload normals_knn16 $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('uniform, reference-order fit forced', l['ms_per_step'])"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_algorithms.py -q -m gpu -k "knn or normals" 2>&1 | grep -E "passed|failed"
timeout 600 python tools/fuzz_knn_sparse.py 40 701 2>&1 | grep -v amdgpu.ids | tail -1
