cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python tools/abab.py --workload normals_knn16_sheet --a "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_r5fit.so" --b "X=1" --pairs 6 --steps 4 --out gpurun_out/r06/abab_sheet_fit.txt 2>&1 | tail -8
