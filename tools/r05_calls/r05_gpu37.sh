cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python tools/abab.py --workload normals_knn16 --pairs 6 --steps 5 --a "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_literals.so" --b "" --out gpurun_out/r05/abab_const_table.txt | tail -6
python tools/abab.py --workload normals_knn16_sheet --pairs 4 --steps 5 --a "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_literals.so" --b "" --out gpurun_out/r05/abab_const_table_sheet.txt | tail -4
timeout 170 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "structured_volume or one_pass or quantised or cross_lane" 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
