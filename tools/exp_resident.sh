#!/bin/bash
# Same-box sweep of the blocks-resident-per-CU cap (PST_RESIDENT, kernels.hpp lds_with_resident_cap) over the record-side workloads:
# one bench line per (workload, cap); prints workload, cap, roofline.frac, kernel ms and the plan family.
# usage: tools/exp_resident.sh "<workloads>" "<caps>" [out.jsonl]
WL=${1:-"las0_to_columns rawlas_to_columns columns_to_las0 las0_encode rawlas_to_records filter_las0_columnar filter_las0_interleaved filter_big_interleaved benchlayout_records_to_columns columns_to_custom41"}
CAPS=${2:-"0 8 6 5 4 3 2"}
OUT=${3:-gpurun_out/r05/resident_sweep.jsonl}
mkdir -p $(dirname $OUT)
for w in $WL; do
  for c in $CAPS; do
    line=$(PST_RESIDENT=$c timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-north-star 2>/dev/null | tail -1)
    echo "{\"resident\": $c, \"line\": $line}" >> $OUT
    echo "$line" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-34s resident=%s frac=%.4f kernel_ms=%.4f min=%.4f plan=%s' % ('$w', '$c', d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['config'].get('plan')))
"
  done
done
