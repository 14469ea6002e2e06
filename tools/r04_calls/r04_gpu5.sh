cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_slices_centroid_views.py tests/test_distributed_gloo.py -m gpu -x -q 2>&1 | tail -8
( time timeout 1500 python -m pytest tests/test_deep_fuzz.py -m gpu -x -q --durations=8 2>&1 | tail -16 ) 2>&1 | tail -22
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "normal or knn or zz_" 2>&1 | tail -4
cat gpurun_out/curvature_floor_use.json
