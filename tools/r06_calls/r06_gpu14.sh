cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu -k "voxel or normals or knn or sparse or hash or morton" 2>&1 | tail -6
B="--steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-extra-legs"
for w in voxelgrid_xyz normals_knn16 normals_knn16_sheet; do python bench.py --workload $w $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', l['ms_per_step'])"; done
