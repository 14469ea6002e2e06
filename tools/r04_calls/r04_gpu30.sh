cd $GRAFT_REPO_ROOT
run() { python bench.py --workload normals_knn16 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "default: $(run) $(run)"
for m in 22 24 26 30 32 36; do echo "TAU_M=$m: $(PST_KNN_TAU_M=$m run)"; done
for f in 32 40 56; do echo "FLUSH_AT=$f: $(PST_KNN_FLUSH_AT=$f run)"; done
for rx in 3 5 6; do echo "RX=$rx: $(PST_KNN_RX=$rx run)"; done
echo "default: $(run)"
