cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export ROUND=r06 COMMIT=dc84b9c
mkdir -p gpurun_out/r06
B="--no-cpu-baseline --no-north-star --no-extra-legs"
cat > /tmp/specs.txt <<S
normals_knn16 knn_tile2_kernel
normals_knn16_sheet knn_tile2_kernel
voxelgrid_xyz_async voxel_reduce_kernel
S
SPECS_FILE=/tmp/specs.txt bash tools/run_profiles.sh 2>&1 | tail -30
for w in normals_knn16 normals_knn16_sheet; do
echo "== stats lib, $w, 8e6"
PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so timeout 300 python bench.py --workload $w --points 8000000 --steps 1 --warmup 0 $B 2>&1 | grep -a "pst knn tile2\|fault" | tail -4 | cut -c1-900 | tee gpurun_out/r06/knn_stats_$w.txt
done
echo "== bench line"
timeout 900 python bench.py > gpurun_out/r06/bench_line_box1.json 2> gpurun_out/r06/bench_line_box1.err; echo rc=$?; cut -c1-1500 gpurun_out/r06/bench_line_box1.json
