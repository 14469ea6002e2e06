cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | cut -c1-1500
cp gpurun_out/curvature_floor_use.json gpurun_out/r05/r05_curvature_floor_use.json 2>/dev/null
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/r05_bench_line.json; cut -c1-3000 gpurun_out/r05/r05_bench_line.json
timeout 900 python tools/exp_knn_guard.py > gpurun_out/r05/r05_knn_fit_guard.txt 2>&1; grep -vE "^\[pst|amdgpu.ids" gpurun_out/r05/r05_knn_fit_guard.txt | tail -40 | cut -c1-300
bash tools/r05_knn_phases.sh normals_knn16
bash tools/r05_knn_phases.sh normals_knn16_sheet
