cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_algorithms.py tests/test_buffer_converter.py tests/test_las_golden.py tests/test_las_encode.py tests/test_slices_centroid_views.py tests/test_bench_line.py tests/test_distributed_gloo.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -1
for w in convert_affine_bounds bounds; do
  timeout 900 python tools/abab.py --workload $w --a "PST_FOLD_ONE_LAUNCH=0" --b "PST_FOLD_ONE_LAUNCH=1" --pairs 6 --steps 30 --out gpurun_out/r06/abab_fold_one_launch_$w.txt 2>&1 | tail -3
done
