#!/bin/bash
# SQ counters of one kernel of a bench workload (run on the GPU box): tools/pmc_kernel.sh <workload> <kernel-name-substring> [lds]
# ("lds": only the LDS set -- SQ_LDS_UNALIGNED_STALL is the counter that found the unaligned LDS accesses)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
w=$1; k=$2
sets=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES")
[ "$3" = lds ] && sets=("SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS")
for set in "${sets[@]}"; do
  out=gpurun_out/pmc_kernel/$(echo $set | tr ' ' '_')
  rm -rf $out; mkdir -p $out
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $out -o p -- python bench.py --no-cpu-baseline --workload $w --steps 3 --warmup 1 > $out/log.txt 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$out/*_results.db")
if db:
    cur = sqlite3.connect(db[0]).cursor()
    for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%$k%' group by kernel_name, counter_name"):
        print(r[0].split('(')[0][-60:], r[1], f"{r[2]:.4g}", r[3])
PY
done
