#!/bin/bash
# The command set behind profiles/r01d_* (run on the GPU box through gpurun): rocprof passes for the workloads whose kernels changed
# late in round 1 (summarised on the box: the raw databases exceed what gpurun copies back), then one bench line per workload.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof_summ
cp profiles/hbm_traffic.json gpurun_out/prof_summ/hbm_traffic.json
for spec in "convert_affine_bounds vec3f64_stream_kernel" "filter_big_interleaved filter_scatter_kernel" "filter_big_columnar filter_scatter_kernel" \
            "columns_to_custom41 convert_tile_kernel" "las1_records_to_custom27 convert_tile_kernel" "normals_knn16 knn_grid_kernel" "voxelgrid_xyz voxel_reduce_kernel"; do
  set -- $spec
  tools/profile_round.sh $1 > /dev/null 2>&1
  python tools/rocprof_summary.py --round r01d --workload $1 --kernel $2 --out gpurun_out/prof_summ --kt gpurun_out/prof/$1/kt/bench_results.db \
    --fetch gpurun_out/prof/$1/fetch/bench_results.db --write gpurun_out/prof/$1/write/bench_results.db --cmd "python bench.py --no-cpu-baseline --workload $1" > /dev/null
  rm -rf gpurun_out/prof/$1
done
rm -f gpurun_out/r01d_workloads.jsonl
for w in convert_affine_bounds bounds las0_to_columns las0_to_columns_bounds rawlas_to_columns rawlas_to_columns_bounds rawlas_to_records columns_to_las0 columns_to_custom41 las1_records_to_custom27 las0_encode filter_big_columnar filter_big_interleaved voxelgrid_xyz narrow_f64_f32 normals_knn16; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r01d_workloads.jsonl
done
wc -l gpurun_out/r01d_workloads.jsonl; ls gpurun_out/prof_summ
