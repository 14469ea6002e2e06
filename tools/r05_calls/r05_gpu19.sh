cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
echo "== knn stats (8e6 points: the stats build's counters are contended atomics)"
PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so timeout 400 python bench.py --workload normals_knn16 --points 8000000 --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-extra-legs > gpurun_out/r05/knn_stats_raw.txt 2>&1
echo "rc=$?"; grep -a "pst knn" gpurun_out/r05/knn_stats_raw.txt | tail -6 | cut -c1-700; tail -c 600 gpurun_out/r05/knn_stats_raw.txt
PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so timeout 400 python bench.py --workload normals_knn16_sheet --points 8000000 --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-extra-legs > gpurun_out/r05/knn_stats_sheet_raw.txt 2>&1
echo "rc=$?"; grep -a "pst knn" gpurun_out/r05/knn_stats_sheet_raw.txt | tail -6 | cut -c1-700
echo "== jit test"
timeout 900 python -m pytest tests/test_expressions.py tests/test_jit.py -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids" | tail -6 | cut -c1-600
echo "== lines"
RANDOM_SEEDS="1" timeout 1000 bash tools/r05_lines.sh boxB 2>&1 | tail -64 | cut -c1-160
