#!/bin/bash
# Per-kernel times of an arbitrary command (run on the GPU box): tools/kt_cmd.sh <cmd...>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=/tmp/kt_cmd
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o t -- "$@" > $out/log.txt 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$out/**/t_results.db", recursive=True)[0]
for r in sqlite3.connect(db).execute("select name,total_calls,average,percentage from top_kernels limit 16"):
    print(f"{r[0][:100]:100s} {r[1]:5d} {r[2]/1000:10.2f} ms {r[3]:6.1f} %")
PY
tail -2 $out/log.txt
