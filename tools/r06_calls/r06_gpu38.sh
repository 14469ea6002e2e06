cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for w in rawlas_to_columns rawlas_to_columns_bounds; do
timeout 900 python tools/abab.py --workload $w --a "X=1" --b "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_dec5.so" --pairs 6 --steps 20 --out gpurun_out/r06/abab_decode_waves5_$w.txt 2>&1 | tail -3
done
