cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_voxel_grid.py -x -q -m gpu -k "voxel or normals or knn" 2>&1 | tail -4
for rep in 1 2; do for s in own rocprim; do for w in voxelgrid_xyz normals_knn16; do
  PST_SORT=$s python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w sort=$s', d['ms_per_step'])"
done; done; done
