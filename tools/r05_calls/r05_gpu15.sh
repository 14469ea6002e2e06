cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | cut -c1-1500
cp gpurun_out/curvature_floor_use.json gpurun_out/r05/ 2>/dev/null; cat gpurun_out/curvature_floor_use.json 2>/dev/null
python tools/abab.py --workload normals_knn16 --steps 6 --a "PST_KNN_FIT_GUARD=0" --b "" --out gpurun_out/r05/abab_4.txt
timeout 900 python tools/exp_knn_guard.py > gpurun_out/r05/knn_guard.txt 2>&1; grep -vE "^\[pst|amdgpu.ids" gpurun_out/r05/knn_guard.txt | cut -c1-500
bash tools/r05_knn_phases.sh normals_knn16
