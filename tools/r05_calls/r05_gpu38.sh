cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python tools/abab.py --workload normals_knn16 --pairs 6 --steps 5 --a "" --b "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_fitb8.so" --out gpurun_out/r05/abab_fitb8_noscratch.txt | tail -6
python tools/abab.py --workload normals_knn16_sheet --pairs 3 --steps 5 --a "" --b "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_fitb8.so" --out gpurun_out/r05/abab_fitb8_noscratch_sheet.txt | tail -4
