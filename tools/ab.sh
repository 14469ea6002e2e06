#!/bin/bash
# A/B of two library builds on the SAME box: tools/ab.sh <old.so> <new.so> <workload> [<workload> ...]
old=$1; new=$2; shift 2
for w in "$@"; do
  for rep in 1 2; do
    for lib in "$old" "$new"; do
      PASTURE_AMD_LIB=$PWD/$lib python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$lib', d['roofline']['achieved'], d['roofline']['kernel_ms_avg'])"
    done
  done
done
