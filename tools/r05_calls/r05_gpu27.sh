cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-north-star --no-extra-legs"
for w in normals_knn16 normals_knn16_sheet; do
echo "== stats lib (counters on lane 1), $w, 8e6"
PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so timeout 300 python bench.py --workload $w --points 8000000 --steps 1 --warmup 0 $B 2>&1 | grep -a "pst knn tile2\|fault" | tail -3 | cut -c1-700 | tee -a gpurun_out/r05/knn_stats_$w.txt
done
