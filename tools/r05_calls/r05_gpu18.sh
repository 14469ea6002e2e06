cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
echo "== full suite"
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids" | tail -25 | cut -c1-1200
cp gpurun_out/curvature_floor_use.json gpurun_out/r05/r05_curvature_floor_use.json 2>/dev/null
echo "== bench line"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/r05_bench_line.json; cut -c1-600 gpurun_out/r05/r05_bench_line.json
echo "== knn stats"
PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so timeout 600 python bench.py --workload normals_knn16 --steps 1 --warmup 1 --no-cpu-baseline --no-north-star --no-extra-legs 2>&1 | grep "pst knn" | tail -4 | tee gpurun_out/r05/knn_stats.txt
PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so timeout 600 python bench.py --workload normals_knn16_sheet --steps 1 --warmup 1 --no-cpu-baseline --no-north-star --no-extra-legs 2>&1 | grep "pst knn" | tail -4 | tee gpurun_out/r05/knn_stats_sheet.txt
echo "== profiles"
SPECS_FILE=tools/r05_specs_final.txt timeout 1500 bash tools/run_profiles_r05.sh 2>&1 | tail -25 | cut -c1-300
echo "== lines"
RANDOM_SEEDS="1" timeout 1200 bash tools/r05_lines.sh boxA 2>&1 | tail -70 | cut -c1-200
