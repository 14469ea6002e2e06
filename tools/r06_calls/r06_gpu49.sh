cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "2_pow_32_typed_las" -rs 2>&1 | tail -25 | cut -c1-600 | tee gpurun_out/r06/beyond_2pow32_c.txt
