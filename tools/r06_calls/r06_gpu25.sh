cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 3300 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tee gpurun_out/r06/gpu_suite_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/r06/gpu_suite_final.txt
timeout 900 python tools/fuzz_knn_sparse.py 60 601 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300 | tee -a gpurun_out/r06/gpu_suite_final.txt
