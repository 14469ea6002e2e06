cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_voxel_grid.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
for w in voxelgrid_xyz voxelgrid_xyz_async; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms_avg'), 'frac', d['roofline']['frac'])"
done
done
# the same with the host's cores kept busy (the synchronous call waits for the host three times per call)
python - <<'PY' &
import time
t=time.time()
while time.time()-t < 60: pass
PY
for i in $(seq 1 $(( $(nproc) - 1 ))); do (python -c "
import time
t=time.time()
while time.time()-t < 45: pass" &) ; done
sleep 2
for w in voxelgrid_xyz voxelgrid_xyz_async; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BUSY HOST $w', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms_avg'))"
done
wait
