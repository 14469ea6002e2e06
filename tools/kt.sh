#!/bin/bash
# Per-kernel times of one bench workload (run on the GPU box): tools/kt.sh <workload> [extra bench args]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
w=$1; shift
out=/tmp/kt_$w
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o bench -- python bench.py --no-cpu-baseline --workload $w --steps 4 --warmup 1 "$@" > $out/bench.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$out/**/bench_results.db", recursive=True)[0]
for r in sqlite3.connect(db).execute("select name,total_calls,average,percentage from top_kernels limit 14"):
    print(f"{r[0][:100]:100s} {r[1]:5d} {r[2]/1000:10.1f} us {r[3]:6.1f} %")
PY
grep -E "pst knn" $out/bench.log | head -2
