cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
( time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r04/gputest_full.log 2>&1
cat gpurun_out/r04/gputest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
# busy host: synchronous against stream-ordered kNN
for i in $(seq 1 $(( $(nproc) - 1 ))); do (python -c "
import time
t=time.time()
while time.time()-t < 50: pass" &) ; done
sleep 2
for w in normals_knn16 normals_knn16_async; do
  python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BUSY HOST $w', 'ms_per_step', d['ms_per_step'])"
done
wait
