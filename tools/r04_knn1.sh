#!/bin/bash
# round 4, kNN plane fit in one pass: tests, same-box A/B against the reference-order instance, kernel trace + WRITE_SIZE / FETCH_SIZE of the box kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests -m gpu -x -q -k "normal or knn" 2>&1 | tail -6
for rep in 1 2; do
for fit in pivot seq; do
  for w in normals_knn16 normals_knn16_sheet; do
    PST_KNN_FIT=$fit python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w fit=$fit ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms_avg'))"
  done
done
done
for fit in pivot seq; do
  d=/tmp/kt_$fit; rm -rf $d; mkdir -p $d
  PST_KNN_FIT=$fit timeout 600 rocprofv3 --kernel-trace --stats -d $d -o p -- python bench.py --no-cpu-baseline --workload normals_knn16 --steps 3 --warmup 1 > $d/log.txt 2>&1
  python - <<PY
import sqlite3, glob, re
db = glob.glob("$d/*_results.db")
cur = sqlite3.connect(db[0]).cursor()
print("== fit=$fit")
for r in list(cur.execute("select name, total_calls, average, percentage from top_kernels"))[:8]:
    nm = re.sub(r"\(anonymous namespace\)::|pstk::|pstn::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", r[0])
    print(f"  {nm[:70]:70s} calls {r[1]:3d} avg_us {r[2]:10.1f} pct {r[3]:5.1f}")
PY
  for set in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
    d=/tmp/pmc_$fit; rm -rf $d; mkdir -p $d
    PST_KNN_FIT=$fit timeout 600 rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python bench.py --no-cpu-baseline --workload normals_knn16 --steps 2 --warmup 1 > $d/log.txt 2>&1
    python - <<PY
import sqlite3, glob
db = glob.glob("$d/*_results.db")
if db:
    cur = sqlite3.connect(db[0]).cursor()
    for r in cur.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%knn_tile2%' group by counter_name"):
        print(f"  {r[0]:28s} {r[1]:.6g}  ({r[2]})")
PY
  done
done
