cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_filter_append.py tests/test_jit.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -1
timeout 900 python tools/abab.py --workload filter_las0_columnar --a "PST_FILTER_THREE_BLOCKS=0" --b "PST_FILTER_THREE_BLOCKS=1" --pairs 6 --steps 20 --out gpurun_out/r06/abab_filter_three_blocks.txt 2>&1 | tail -3
timeout 900 python tools/abab.py --workload las0_encode --a "X=1" --b "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_enc4.so" --pairs 6 --steps 20 --out gpurun_out/r06/abab_encode_waves4.txt 2>&1 | tail -3
