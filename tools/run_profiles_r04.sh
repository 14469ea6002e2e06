#!/bin/bash
# The command set behind profiles/r04_* (run on the GPU box through gpurun): per workload a `rocprofv3 --kernel-trace --stats` pass and separate
# `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py, summarised ON the box (tools/rocprof_summary.py).
# SPECS entries: "<workload> <dominant kernel substring> [extra bench.py flags...]"; the profile key is workload[_seedN][_plan].
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r04/prof_summ
mkdir -p $out
cp profiles/hbm_traffic.json $out/hbm_traffic.json
[ -f profiles/knn_valu.json ] && cp profiles/knn_valu.json $out/knn_valu.json
DEFAULT_SPECS=(
 "convert_affine_bounds vec3f64_stream_kernel"
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
  case "$extra" in *"--plan interpreted"*) key=${key}_interpreted;; *"--plan specialised"*) key=${key}_specialised;; esac
  for pass in kt fetch write; do
    o=gpurun_out/prof/$key/$pass; mkdir -p "$o"
    case $pass in
      kt) args="--kernel-trace --stats" ;;
      fetch) args="--kernel-trace --pmc FETCH_SIZE" ;;
      write) args="--kernel-trace --pmc WRITE_SIZE" ;;
    esac
    steps=5; [ "$pass" != kt ] && steps=3
    timeout 600 rocprofv3 $args -d "$o" -o bench -- python bench.py --no-cpu-baseline --no-north-star --workload "$w" $extra --steps $steps --warmup 1 > "$o/bench.log" 2>&1
    echo "$key $pass rc=$? $(tail -c 200 "$o/bench.log" | tr '\n' ' ')"
  done
  python tools/rocprof_summary.py --round r04 --workload $w --key $key --kernel "$kern" --out $out --kt gpurun_out/prof/$key/kt/bench_results.db \
    --fetch gpurun_out/prof/$key/fetch/bench_results.db --write gpurun_out/prof/$key/write/bench_results.db --cmd "python bench.py --no-cpu-baseline --workload $w $extra" > /dev/null
  rm -rf gpurun_out/prof/$key
done
ls $out
