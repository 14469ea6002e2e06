cd $GRAFT_REPO_ROOT
B="--workload normals_knn16 --steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-extra-legs"
run() { env "$@" python bench.py $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'])"; }
for i in 1 2 3; do echo "batch 4: $(run PST_KNN_BATCH=4)   batch 8: $(run PST_KNN_BATCH=8)"; done
PST_KNN_BATCH=8 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or normals" 2>&1 | tail -3
