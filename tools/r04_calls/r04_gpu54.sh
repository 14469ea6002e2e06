cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "filter" 2>&1 | tail -2
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"; }
for rep in 1 2 3; do for w in filter_las0_interleaved filter_big_columnar; do for v in 0 1; do
  PST_FILTER_SCAN_BLOCKS=$v python bench.py --no-cpu-baseline --no-north-star --workload $w --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | line "$w SCAN_BLOCKS=$v"
done; done; done
d=gpurun_out/prof/scan; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --stats -d $d -o bench -- python bench.py --no-cpu-baseline --no-north-star --workload filter_las0_interleaved --steps 10 --warmup 2 > $d/bench.log 2>&1
python - <<'PY'
import sqlite3
cur = sqlite3.connect("gpurun_out/prof/scan/bench_results.db").cursor()
for r in cur.execute("select name, average from top_kernels where name like '%tile_scan%' or name like '%mask_count%' or name like '%filter_stream%'"): print(r[0][:80], r[1])
PY
rm -rf gpurun_out/prof/scan
