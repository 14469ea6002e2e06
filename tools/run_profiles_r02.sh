#!/bin/bash
# The command set behind profiles/r02_* (run on the GPU box through gpurun): per workload a `rocprofv3 --kernel-trace --stats` pass and separate
# `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py (tools/profile_round.sh), summarised ON the box (the raw databases exceed what gpurun
# copies back) with the per-access-pattern fetch factors of tools/pmc_calibrate; then one bench line per workload from the same box.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r02/prof_summ
mkdir -p $out
cp profiles/hbm_traffic.json $out/hbm_traffic.json
tools/pmc_calibrate.sh > /dev/null 2>&1
cp gpurun_out/r02/pmc_calibration.json $out/pmc_calibration.json
for spec in "convert_affine_bounds vec3f64_stream_kernel" "normals_knn16 knn_tile_kernel" "voxelgrid_xyz voxel_reduce_kernel" "las0_to_columns las_records_to_columns_kernel" \
            "filter_big_interleaved filter_big_records_kernel" "filter_big_columnar filter_scatter_kernel" "columns_to_custom41 convert_tile_static_kernel" \
            "las1_records_to_custom27 convert_tile_static_kernel" "benchlayout_records_to_records convert_tile_static_kernel" "las0_encode las_encode_kernel" \
            "rawlas_to_columns las_decode_kernel" "normals_knn16_sheet knn_tile_kernel"; do
  set -- $spec
  tools/profile_round.sh $1 > /dev/null 2>&1
  python tools/rocprof_summary.py --round r02 --workload $1 --kernel "$2" --out $out --kt gpurun_out/prof/$1/kt/bench_results.db \
    --fetch gpurun_out/prof/$1/fetch/bench_results.db --write gpurun_out/prof/$1/write/bench_results.db --cmd "python bench.py --no-cpu-baseline --workload $1" > /dev/null
  rm -rf gpurun_out/prof/$1
done
rm -f gpurun_out/r02/r02_workloads.jsonl
for w in convert_affine_bounds bounds las0_to_columns las0_to_columns_bounds rawlas_to_columns rawlas_to_columns_bounds rawlas_to_records columns_to_las0 columns_to_custom41 las1_records_to_custom27 benchlayout_records_to_columns benchlayout_columns_to_records benchlayout_records_to_records las0_encode filter_big_columnar filter_big_interleaved voxelgrid_xyz narrow_f64_f32 normals_knn16 normals_knn16_sheet; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r02/r02_workloads.jsonl
done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r02/r02_bench_line.json
wc -l gpurun_out/r02/r02_workloads.jsonl; ls $out
