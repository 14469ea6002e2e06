#!/bin/bash
# One bench line per workload on this box -> gpurun_out/r06/r06_workloads_$1.jsonl.  Generic conversion / compaction workloads run BOTH ways:
# --plan interpreted (PST_JIT=0: the mapping list interpreted by the tile kernels) and --plan specialised (in-tree instantiation or hipRTC);
# config.plan in every line is the kernel family the library reports (pst_last_plan_kinds), not a label of this script.
cd "$(dirname "$0")/.."
out=gpurun_out/r06/r06_workloads_${1:-x}.jsonl; mkdir -p gpurun_out/r06; rm -f $out
GENERIC="las0_to_columns las0_to_columns_bounds rawlas_to_columns rawlas_to_columns_bounds rawlas_to_records columns_to_las0 columns_to_custom41 las1_records_to_custom27 benchlayout_records_to_columns benchlayout_columns_to_records benchlayout_records_to_records"
for w in $GENERIC; do
  for plan in interpreted specialised; do
    python bench.py --workload $w --plan $plan --steps 10 --warmup 2 --no-cpu-baseline --no-north-star --no-extra-legs 2>/dev/null | tail -1 >> $out
  done
done
for seed in ${RANDOM_SEEDS:-1 4 7}; do
  for w in randomlayout_records_to_columns randomlayout_columns_to_records randomlayout_records_to_records; do
    for plan in interpreted specialised; do
      python bench.py --workload $w --layout-seed $seed --plan $plan --steps 10 --warmup 2 --no-cpu-baseline --no-north-star --no-extra-legs 2>/dev/null | tail -1 >> $out
    done
  done
done
for w in filter_big_columnar filter_big_interleaved filter_las0_columnar filter_las0_interleaved filter_las3_columnar filter_las3_interleaved filter_las9_interleaved; do
  for plan in interpreted specialised; do   # (interpreted = the gather kernels; specialised = the streaming kernels, in-tree or compiled before the timed region)
    python bench.py --workload $w --plan $plan --steps 10 --warmup 2 --no-cpu-baseline --no-north-star --no-extra-legs 2>/dev/null | tail -1 >> $out
  done
done
if [ -z "${GENERIC_ONLY:-}" ]; then
for w in convert_affine_bounds bounds las0_encode voxelgrid_xyz voxelgrid_xyz_async narrow_f64_f32 normals_knn16 normals_knn16_async normals_knn16_sheet; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-north-star --no-extra-legs 2>/dev/null | tail -1 >> $out
done
fi
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d['config']
    print(c['workload'].split(':')[0][:34].ljust(34), str(c.get('plan_requested')).ljust(12), str(c.get('plan')).ljust(34), d['ms_per_step'], d['roofline']['frac'])
PY
