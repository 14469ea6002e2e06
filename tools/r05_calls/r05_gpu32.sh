cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python tools/abab.py --workload normals_knn16 --pairs 5 --steps 5 --a "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_fitserial.so" --b "" --out gpurun_out/r05/abab_fitbatch8.txt | tail -6
python tools/abab.py --workload normals_knn16 --pairs 4 --steps 5 --a "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_fitb4.so" --b "" --out gpurun_out/r05/abab_fitbatch4v8.txt | tail -6
python tools/abab.py --workload normals_knn16_sheet --pairs 3 --steps 5 --a "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_fitserial.so" --b "" --out gpurun_out/r05/abab_fitbatch8_sheet.txt | tail -4
