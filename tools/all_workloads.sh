rm -f gpurun_out/r01d_workloads_box2.jsonl
for w in convert_affine_bounds bounds las0_to_columns las0_to_columns_bounds rawlas_to_columns rawlas_to_columns_bounds rawlas_to_records columns_to_las0 columns_to_custom41 las1_records_to_custom27 las0_encode filter_big_columnar filter_big_interleaved voxelgrid_xyz narrow_f64_f32 normals_knn16; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r01d_workloads_box2.jsonl
done
python - <<PY
import json
for l in open('gpurun_out/r01d_workloads_box2.jsonl'):
    d=json.loads(l); print(d['config']['workload'].split(':')[0][:28].ljust(28), d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
PY
