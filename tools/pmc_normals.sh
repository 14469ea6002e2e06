#!/bin/bash
# SQ counters of the kNN kernel (run on the GPU box): tools/pmc_normals.sh <points>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
n=${1:-10000000}
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  out=gpurun_out/pmc_normals/$(echo $set | tr ' ' '_')
  mkdir -p $out
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $out -o p -- python tools/exp_normals.py $n > $out/log.txt 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$out/*_results.db")[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%knn_grid%' group by kernel_name, counter_name"):
    print(r[0][:40], r[1], f"{r[2]:.4g}", r[3])
PY
done
