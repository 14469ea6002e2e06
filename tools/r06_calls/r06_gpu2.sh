cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-north-star --no-extra-legs --workload normals_knn16 --steps 5 --warmup 1"
for m in 0 1 2 0 1 2; do
echo "== overlap mode $m"; PST_KNN_EXP_OVERLAP=$m timeout 300 python bench.py $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'])"
done
