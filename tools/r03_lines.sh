#!/bin/bash
# One bench line per workload on this box -> gpurun_out/r03/r03_workloads_$1.jsonl (no profiling passes)
cd "$(dirname "$0")/.."
out=gpurun_out/r03/r03_workloads_${1:-x}.jsonl; mkdir -p gpurun_out/r03; rm -f $out
for w in convert_affine_bounds bounds las0_to_columns las0_to_columns_bounds rawlas_to_columns rawlas_to_columns_bounds rawlas_to_records columns_to_las0 columns_to_custom41 las1_records_to_custom27 benchlayout_records_to_columns benchlayout_columns_to_records benchlayout_records_to_records las0_encode filter_big_columnar filter_big_interleaved voxelgrid_xyz narrow_f64_f32 normals_knn16 normals_knn16_sheet; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $out
done
wc -l $out
