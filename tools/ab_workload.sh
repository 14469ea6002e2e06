#!/bin/bash
# Same-box A/B of library builds on one bench workload (run on the GPU box): tools/ab_workload.sh <workload> <lib.so> ...   ("-" = the in-tree build)
cd "$(dirname "$0")/.."
w=$1; shift
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset PASTURE_AMD_LIB; else export PASTURE_AMD_LIB=$lib; fi
  r=$(python bench.py --workload $w --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])")
  echo "== $w $lib: $r"
done
