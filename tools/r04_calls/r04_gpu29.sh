cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
d=gpurun_out/prof/scan; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --stats -d $d -o bench -- python bench.py --no-cpu-baseline --no-north-star --workload filter_big_columnar --steps 10 --warmup 2 > $d/bench.log 2>&1
python - <<'PY'
import sqlite3
cur = sqlite3.connect("gpurun_out/prof/scan/bench_results.db").cursor()
for r in cur.execute("select name, average from top_kernels where name like '%tile_scan%' or name like '%mask_count%' or name like '%filter_big%'"): print(r)
PY
rm -rf gpurun_out/prof/scan
