#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE PMC passes for the bench workloads.
# Usage: tools/profile_round.sh <workload> [<workload> ...]   -> gpurun_out/prof/<workload>/{kt,fetch,write}/bench_results.db
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for w in "$@"; do
  for pass in kt fetch write; do
    out=gpurun_out/prof/$w/$pass
    mkdir -p "$out"
    case $pass in
      kt) args="--kernel-trace --stats" ;;
      fetch) args="--kernel-trace --pmc FETCH_SIZE" ;;
      write) args="--kernel-trace --pmc WRITE_SIZE" ;;
    esac
    steps=5; [ "$pass" != kt ] && steps=3
    timeout 600 rocprofv3 $args -d "$out" -o bench -- python bench.py --no-cpu-baseline --no-north-star --workload "$w" --steps $steps --warmup 1 > "$out/bench.log" 2>&1
    echo "$w $pass rc=$? $(tail -c 300 "$out/bench.log" | tr '\n' ' ' | cut -c1-200)"
  done
done
