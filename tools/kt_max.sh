#!/bin/bash
# Longest single dispatch per kernel + longest gaps between consecutive dispatches of a command (run on the GPU box): tools/kt_max.sh <cmd...>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=/tmp/kt_max
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o t -- "$@" > $out/log.txt 2>&1
tail -1 $out/log.txt | cut -c1-300
python - <<PY
import sqlite3, glob
db = glob.glob("$out/**/t_results.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if 'kernel' in t.lower() and 'dispatch' in t.lower()] or [t for t in tabs if t == 'kernels']
print("tables:", kt[:5])
v = 'kernels' if 'kernels' in tabs else kt[0]
cols = [r[1] for r in con.execute(f"pragma table_info({v})")]
print(cols)
rows = list(con.execute(f"select name, start, end from {v} order by start"))
import collections
mx = collections.defaultdict(float)
for n, s, e in rows: mx[n] = max(mx[n], (e - s) / 1e6)
for n, d in sorted(mx.items(), key=lambda x: -x[1])[:6]: print(f"{d:10.2f} ms max  {n[:90]}")
gaps = sorted(((rows[i + 1][1] - rows[i][2]) / 1e6, rows[i][0][:50], rows[i + 1][0][:50]) for i in range(len(rows) - 1))[-5:]
for g in gaps: print(f"gap {g[0]:10.2f} ms after {g[1]} before {g[2]}")
PY
