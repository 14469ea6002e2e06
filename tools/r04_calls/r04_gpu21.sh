cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
d=/tmp/kt; rm -rf $d; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --stats -d $d -o p -- python bench.py --no-cpu-baseline --workload normals_knn16_sheet --steps 3 --warmup 1 > $d/log.txt 2>&1
python - <<PY
import sqlite3, glob, re
db = glob.glob("$d/*_results.db")
cur = sqlite3.connect(db[0]).cursor()
for r in list(cur.execute("select name, total_calls, average, percentage from top_kernels"))[:16]:
    nm = re.sub(r"\(anonymous namespace\)::|pstk::|pstn::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", r[0])
    print(f"  {nm[:70]:70s} calls {r[1]:3d} avg_us {r[2]:10.1f} pct {r[3]:5.1f}")
PY
