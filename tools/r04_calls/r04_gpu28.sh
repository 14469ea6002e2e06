cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "filter or static or voxel" 2>&1 | tail -2
d=gpurun_out/prof/scan; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --stats -d $d -o bench -- python bench.py --no-cpu-baseline --no-north-star --workload filter_big_columnar --steps 10 --warmup 2 > $d/bench.log 2>&1
python - <<'PY'
import sqlite3
cur = sqlite3.connect("gpurun_out/prof/scan/bench_results.db").cursor()
for r in cur.execute("select name, calls, average from top_kernels where name like '%tile_scan%' or name like '%mask_count%' or name like '%filter_big%'"): print(r)
PY
rm -rf gpurun_out/prof/scan
for rep in 1 2 3; do for w in filter_big_columnar filter_big_interleaved; do
  python bench.py --no-cpu-baseline --no-north-star --workload $w --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"
done; done
