#!/bin/bash
# Round 6, review item 6: does the phase of the columns' base addresses explain the box-to-box spread of the ten-column kernels?  Every column of a
# HashMapBuffer is its own allocation (same phase of the channel interleave); PST_COLUMN_STAGGER places column a `a x step` bytes into its allocation.
# NOTE: PST_COLUMN_STAGGER was read by an allocator patch (buffer.cpp: column a placed (a mod 16) x step bytes into its padded allocation) that was REMOVED after this
# sweep showed no effect (profiles/r06_column_stagger_sweep.txt); the script is the record of how the sweep was run.
# Sweeps the step for the workloads that walk many columns, kernel ms (HIP events, 20 steps) per step; run on several boxes (each gpurun call is a fresh one).
cd "$(dirname "$0")/.."
B="--steps 20 --warmup 3 --no-cpu-baseline --no-north-star --no-extra-legs"
run() { PST_COLUMN_STAGGER=$1 python bench.py --workload $2 $B 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % l['roofline']['kernel_ms_avg'])"; }
steps="0 256 512 1024 2048 4096 4352 8192 12544 65536 0"
printf "%-26s" "workload \\ step"; for s in $steps; do printf "%8s" $s; done; echo
for w in las0_to_columns rawlas_to_columns columns_to_las0 las0_encode filter_las0_columnar convert_affine_bounds; do
  printf "%-26s" $w; for s in $steps; do printf "%8s" "$(run $s $w)"; done; echo
done
