#!/bin/bash
# Per-phase budget of the kNN box kernel (round-4 review, item 3): knn_tile2_kernel's duration and vector (wave64) instruction count with one phase
# switched off at a time (PST_KNN_ABLATE: 16 = no staging copy, 4 = no scan, 8 = nothing queued, 1 = no insertion, 2 = no plane fit, 32 = no proof
# results / fit / stores), each from its own `rocprofv3 --kernel-trace --stats` and `--pmc SQ_INSTS_VALU SQ_WAVES` pass of the 10^8-point workload,
# plus the wave-clock split of the -DPST_KNN_STATS build.  An ablated run computes wrong results: this is a budget, not a benchmark.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-normals_knn16}
out=gpurun_out/r05/knn_phases_$W.txt; mkdir -p gpurun_out/r05; : > $out
echo "# $W: knn_tile2_kernel by phase (tools/r05_knn_phases.sh)" >> $out
PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-north-star --no-extra-legs 2>&1 | grep "pst knn tile2" | tail -2 >> $out
for ab in 0 16 4 8 1 2 32; do
  for pass in kt pmc; do
    d=gpurun_out/prof/knnph_${ab}_$pass; rm -rf $d; mkdir -p $d
    if [ $pass = kt ]; then args="--kernel-trace --stats"; else args="--kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS"; fi
    PST_KNN_ABLATE=$ab timeout 600 rocprofv3 $args -d $d -o bench -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra-legs > $d/log 2>&1
  done
  python - $ab gpurun_out/prof/knnph_${ab}_kt/bench_results.db gpurun_out/prof/knnph_${ab}_pmc/bench_results.db >> $out <<'PY'
import sqlite3, sys
ab, kt_db, pmc_db = sys.argv[1:4]
kt = sqlite3.connect(kt_db).cursor()
ms = [(r[0], r[1] / 1e3, r[2]) for r in kt.execute("select name, average, total_calls from top_kernels where name like '%knn_tile2_kernel%'")]
pm = sqlite3.connect(pmc_db).cursor()
cnt = {r[0]: r[1] for r in pm.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%knn_tile2_kernel%' group by counter_name")}
allk = [(r[0], r[1] / 1e3, r[2]) for r in kt.execute("select name, average, total_calls from top_kernels order by total_duration desc limit 6")]
name, t, calls = ms[0] if ms else ("?", 0.0, 0)
print(f"ablate={int(ab):2d}  knn_tile2_kernel {t:8.3f} ms  VALU wave instr {cnt.get('SQ_INSTS_VALU', 0):.4e}  SALU {cnt.get('SQ_INSTS_SALU', 0):.3e}  LDS {cnt.get('SQ_INSTS_LDS', 0):.3e}  waves {cnt.get('SQ_WAVES', 0):.3e}")
PY
  rm -rf gpurun_out/prof/knnph_${ab}_kt gpurun_out/prof/knnph_${ab}_pmc
done
cat $out
