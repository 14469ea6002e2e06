cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) 2>&1 | tail -8
python bench.py --workload normals_knn16_sheet --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sheet ms_per_step', d['ms_per_step'])"
