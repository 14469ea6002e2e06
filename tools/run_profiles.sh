#!/bin/bash
export ROUND=${ROUND:-r06}; export COMMIT=${COMMIT:-unknown}
# The command set behind profiles/${ROUND}_* (run on the GPU box through gpurun): per workload a `rocprofv3 --kernel-trace --stats` pass and separate
# `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py, summarised ON the box (tools/rocprof_summary.py).
# SPECS entries: "<workload> <dominant kernel substring> [extra bench.py flags...]"; the profile key is workload[_seedN][_plan].
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/${ROUND}/prof_summ
mkdir -p $out
cp profiles/hbm_traffic.json $out/hbm_traffic.json
[ -f profiles/knn_valu.json ] && cp profiles/knn_valu.json $out/knn_valu.json
DEFAULT_SPECS=(
 "convert_affine_bounds vec3f64_stream2_kernel"
 "randomlayout_records_to_columns pst_jit_convert --plan specialised --layout-seed 1"
 "randomlayout_columns_to_records pst_jit_convert --plan specialised --layout-seed 1"
 "randomlayout_records_to_records pst_jit_convert --plan specialised --layout-seed 1"
 "randomlayout_records_to_records convert_tile_kernel --plan interpreted --layout-seed 1"
 "benchlayout_records_to_records convert_quad --plan specialised"
 "columns_to_custom41 convert_quad --plan specialised"
 "las1_records_to_custom27 convert_quad --plan specialised"
)
if [ -n "${SPECS_FILE:-}" ]; then mapfile -t SPECS_ARR < "$SPECS_FILE"; else SPECS_ARR=("${DEFAULT_SPECS[@]}"); fi
for spec in "${SPECS_ARR[@]}"; do
  set -- $spec
  w=$1; kern=$2; shift 2; extra="$*"
  key=$w
  case "$extra" in *"--layout-seed"*) key=${key}_seed$(echo "$extra" | sed 's/.*--layout-seed \([0-9]*\).*/\1/');; esac
  pts=100000000
  case "$extra" in *"--points"*) pts=$(echo "$extra" | sed 's/.*--points \([0-9]*\).*/\1/'); key=${key}_$(python -c "print('%.0e' % $pts)" | sed 's/+0*//');; esac
  case "$extra" in *"--plan interpreted"*) key=${key}_interpreted;; *"--plan specialised"*) key=${key}_specialised;; esac
  for pass in kt fetch write; do
    o=gpurun_out/prof/$key/$pass; mkdir -p "$o"
    case $pass in
      kt) args="--kernel-trace --stats" ;;
      fetch) args="--kernel-trace --pmc FETCH_SIZE" ;;
      write) args="--kernel-trace --pmc WRITE_SIZE" ;;
    esac
    steps=5; [ "$pass" != kt ] && steps=3
    timeout 600 rocprofv3 $args -d "$o" -o bench -- python bench.py --no-cpu-baseline --no-north-star --no-extra-legs --workload "$w" $extra --steps $steps --warmup 1 > "$o/bench.log" 2>&1
    echo "$key $pass rc=$? $(tail -c 200 "$o/bench.log" | tr '\n' ' ')"
  done
  python tools/rocprof_summary.py --round ${ROUND} --workload $w --points $pts --key $key --kernel "$kern" --out $out --kt gpurun_out/prof/$key/kt/bench_results.db \
    --fetch gpurun_out/prof/$key/fetch/bench_results.db --write gpurun_out/prof/$key/write/bench_results.db --cmd "python bench.py --no-cpu-baseline --workload $w $extra  [tree: $COMMIT]" > /dev/null
  if [[ $w == normals_knn* ]]; then
    d=gpurun_out/prof/$key/valu; mkdir -p $d
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU -d $d -o bench -- python bench.py --no-cpu-baseline --no-north-star --no-extra-legs --workload $w $extra --steps 3 --warmup 1 > $d/bench.log 2>&1
    python - "$w" "$kern" "$d/bench_results.db" "gpurun_out/prof/$key/kt/bench_results.db" "$out/knn_valu.json" <<'PY'
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
                   "kernel_ms": round(ms.get(name, 0.0), 4) or None, "round": os.environ.get("ROUND", "r06"), "commit": os.environ.get("COMMIT", "unknown")}
        for n2, c2, v2, _ in rows:
            if n2 == name and c2 != "SQ_INSTS_VALU": allv[w][c2.lower()] = round(v2)
json.dump(allv, open(path, "w"), indent=1)
print(w, allv.get(w))
PY
  fi
  rm -rf gpurun_out/prof/$key
done
cp $out/knn_valu.json profiles/knn_valu.json 2>/dev/null
cp $out/hbm_traffic.json profiles/hbm_traffic.json 2>/dev/null
ls $out
