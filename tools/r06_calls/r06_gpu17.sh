cd $GRAFT_REPO_ROOT
B="--steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-extra-legs"
run() { env "$@" python bench.py $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'])"; }
echo "== uniform"
echo "default: $(run X=1 --workload normals_knn16)"
run2() { w=$1; shift; env "$@" python bench.py --workload $w $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'])"; }
for m in 24 26 28 30 32 36; do echo "uniform TAU_M=$m: $(run2 normals_knn16 PST_KNN_TAU_M=$m)"; done
for r in 2 3 4 5 6; do echo "uniform RX=$r: $(run2 normals_knn16 PST_KNN_RX=$r)"; done
for f in 32 40 48 56; do echo "uniform FLUSH_AT=$f: $(run2 normals_knn16 PST_KNN_FLUSH_AT=$f)"; done
echo "uniform default: $(run2 normals_knn16 X=1)"
echo "== sheet"
echo "sheet default: $(run2 normals_knn16_sheet X=1)"
for m in 24 28 32; do echo "sheet TAU_M=$m: $(run2 normals_knn16_sheet PST_KNN_TAU_M=$m)"; done
for r in 1 2 3 4; do echo "sheet RX=$r: $(run2 normals_knn16_sheet PST_KNN_RX=$r)"; done
for c in 8 12 20 32; do echo "sheet CELL_BUDGET=$c: $(run2 normals_knn16_sheet PST_KNN_CELL_BUDGET=$c)"; done
echo "sheet default: $(run2 normals_knn16_sheet X=1)"
