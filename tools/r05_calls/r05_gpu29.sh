cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
echo "== tests"; timeout 900 python -m pytest tests/test_las_golden.py tests/test_las_encode.py tests/test_buffer_converter.py -m gpu -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-400
echo "== abab"; python tools/abab.py --workload columns_to_las0 --steps 20 --a "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_unalignedlds.so" --b "" --out gpurun_out/r05/abab_alignedlds.txt | tail -6
