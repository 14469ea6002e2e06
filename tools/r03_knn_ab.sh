#!/bin/bash
# Round 3 kNN experiments on one box: correctness of the second box kernel, then a same-box A/B of its instances against the first form.
cd "$(dirname "$0")/.."
out=gpurun_out/r03/knn; mkdir -p $out
vars="${VARS:-1 A B C D E F}"
if [ -z "$SKIP_TESTS" ]; then
  for v in ${TEST_VARS:-B}; do
    echo "== tests PST_KNN_VAR=$v"
    PST_KNN_VAR=$v timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "normals or knn or sparse" 2>&1 | tail -6
  done
fi
for rep in 1 2; do
  for v in $vars; do
    for w in normals_knn16 ${SHEET:+normals_knn16_sheet}; do
      PST_KNN_VAR=$v python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB $w var=$v', d['ms_per_step'])"
    done
  done
done | tee $out/ab.txt
if [ -f pasture_amd/libpasture_amd_stats.so ]; then
  for v in ${STAT_VARS:-1 B}; do
    PST_KNN_VAR=$v PST_KNN_DEBUG=1 PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so python bench.py --workload normals_knn16 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "pst knn" | tail -4
  done | tee $out/stats.txt
fi
