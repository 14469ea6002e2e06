cd /tmp; export TMPDIR=/tmp
for lib in cnt cnt4 cnt8; do
rm -rf /tmp/kp_$lib
PASTURE_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$lib.so rocprofv3 --kernel-trace --stats -d /tmp/kp_$lib -o k -- python $GRAFT_REPO_ROOT/bench.py --workload filter_big_columnar --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import sqlite3, glob
cur = sqlite3.connect(glob.glob("/tmp/kp_$lib/*_results.db")[0]).cursor()
for r in cur.execute("select name, total_calls, average from top_kernels where name like '%mask_count%' or name like '%tile_scan%' or name like '%filter_scatter%'"): print("$lib", r[0][:60], r[1], round(r[2],1), "ns avg")
PY
done
