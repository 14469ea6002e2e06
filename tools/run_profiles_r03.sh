#!/bin/bash
# The command set behind profiles/r03_* (run on the GPU box through gpurun): per workload a `rocprofv3 --kernel-trace --stats` pass and separate
# `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py (tools/profile_round.sh), summarised ON the box with the per-access-pattern fetch
# factors of tools/pmc_calibrate; an SQ_INSTS_VALU pass for the kNN workloads (-> knn_valu.json: the instruction-issue bound of the bench
# line); then one bench line per workload from the same box.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r03/prof_summ
mkdir -p $out
cp profiles/hbm_traffic.json $out/hbm_traffic.json
[ -f profiles/knn_valu.json ] && cp profiles/knn_valu.json $out/knn_valu.json
for spec in ${SPECS:-"convert_affine_bounds vec3f64_stream_kernel" "normals_knn16 knn_tile2_kernel" "normals_knn16_sheet knn_tile2_kernel" "voxelgrid_xyz voxel_reduce_kernel" \
            "filter_big_interleaved filter_big_records_kernel" "filter_big_columnar filter_scatter_kernel" "las0_to_columns las_records_to_columns_kernel"}; do
  set -- $spec
  tools/profile_round.sh $1 > /dev/null 2>&1
  python tools/rocprof_summary.py --round r03 --workload $1 --kernel "$2" --out $out --kt gpurun_out/prof/$1/kt/bench_results.db \
    --fetch gpurun_out/prof/$1/fetch/bench_results.db --write gpurun_out/prof/$1/write/bench_results.db --cmd "python bench.py --no-cpu-baseline --workload $1" > /dev/null
  if [[ $1 == normals_knn* ]]; then
    d=gpurun_out/prof/$1/valu; mkdir -p $d
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU -d $d -o bench -- python bench.py --no-cpu-baseline --no-north-star --workload $1 --steps 3 --warmup 1 > $d/bench.log 2>&1
    python - "$1" "$2" "$d/bench_results.db" "gpurun_out/prof/$1/kt/bench_results.db" "$out/knn_valu.json" <<'PY'
import json, os, sqlite3, sys
w, kern, pmc_db, kt_db, path = sys.argv[1:6]
cur = sqlite3.connect(pmc_db).cursor()
rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (f"%{kern}%",)))
kt = sqlite3.connect(kt_db).cursor()
ms = {r[0]: r[1] / 1e3 for r in kt.execute("select name, average from top_kernels where name like ?", (f"%{kern}%",))}  # (average is in microseconds)
allv = json.load(open(path)) if os.path.exists(path) else {}
for name, counter, val, cnt in rows:
    if counter == "SQ_INSTS_VALU":
        allv[w] = {"kernel": name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], "valu_wave_instructions_per_launch": round(val), "launches": cnt, "points": 100000000,
                   "kernel_ms": round(ms.get(name, 0.0), 4) or None, "round": "r03"}
        for n2, c2, v2, _ in rows:
            if n2 == name and c2 != "SQ_INSTS_VALU": allv[w][c2.lower()] = round(v2)
json.dump(allv, open(path, "w"), indent=1)
print(w, allv.get(w))
PY
  fi
  rm -rf gpurun_out/prof/$1
done
cp $out/knn_valu.json profiles/knn_valu.json 2>/dev/null
cp $out/hbm_traffic.json profiles/hbm_traffic.json 2>/dev/null
if [ -z "${NO_LINES:-}" ]; then
rm -f gpurun_out/r03/r03_workloads.jsonl
for w in convert_affine_bounds bounds las0_to_columns las0_to_columns_bounds rawlas_to_columns rawlas_to_columns_bounds rawlas_to_records columns_to_las0 columns_to_custom41 las1_records_to_custom27 benchlayout_records_to_columns benchlayout_columns_to_records benchlayout_records_to_records las0_encode filter_big_columnar filter_big_interleaved voxelgrid_xyz narrow_f64_f32 normals_knn16 normals_knn16_sheet; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r03/r03_workloads.jsonl
done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03/r03_bench_line.json
wc -l gpurun_out/r03/r03_workloads.jsonl
fi
ls $out
