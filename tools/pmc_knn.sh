#!/bin/bash
# SQ counters of the kNN box kernel (run on the GPU box): tools/pmc_knn.sh "<counters>" ["<counters>" ...]   -- one rocprofv3 pass per argument
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
i=0
for set in "$@"; do
  out=/tmp/pmc_knn_$i; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python bench.py --no-cpu-baseline --workload normals_knn16 --steps 1 --warmup 0 > $out/log.txt 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/p_counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for path in f:
    for r in csv.DictReader(open(path)):
        if "knn_tile_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc: print(f"{k:32s} {acc[k]/max(n[k],1):18.0f}  (launches {n[k]})")
PY
  i=$((i+1))
done
