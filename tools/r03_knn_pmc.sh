#!/bin/bash
# SQ counters + kernel trace of the box kernel for the variants in $VARS (run on the GPU box)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r03/knn; mkdir -p $out
for v in ${VARS:-1 B D}; do
  export PST_KNN_VAR=$v
  echo "=== var $v"
  d=/tmp/kt_$v; rm -rf $d; mkdir -p $d
  timeout 600 rocprofv3 --kernel-trace --stats -d $d -o p -- python bench.py --no-cpu-baseline --workload ${W:-normals_knn16} --steps 3 --warmup 1 > $d/log.txt 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$d/*_results.db")
cur = sqlite3.connect(db[0]).cursor()
import re
for r in list(cur.execute("select name, total_calls, average, percentage from top_kernels"))[:${TOPN:-8}]:
    nm = re.sub(r"\(anonymous namespace\)::|pstk::|pstn::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", r[0])
    print(f"  {nm[:60]:60s} calls {r[1]:3d} avg_us {r[2]:10.1f} pct {r[3]:5.1f}")
PY
  if [ -z "$NO_PMC" ]; then
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64"; do
    d=/tmp/pmc_$v; rm -rf $d; mkdir -p $d
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python bench.py --no-cpu-baseline --workload ${W:-normals_knn16} --steps 2 --warmup 1 > $d/log.txt 2>&1
    python - <<PY
import sqlite3, glob
db = glob.glob("$d/*_results.db")
if db:
    cur = sqlite3.connect(db[0]).cursor()
    for r in cur.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%knn_tile%' group by counter_name"):
        print(f"  {r[0]:28s} {r[1]:.5g}  ({r[2]})")
PY
  done
  fi
done 2>&1 | tee $out/pmc_${TAG:-x}.txt
