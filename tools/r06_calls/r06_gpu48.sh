cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for m in env sync 0; do
  if [ $m = env ]; then unset PST_JIT; else export PST_JIT=$m; fi
  echo "== PST_JIT=$m"
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "2_pow_32_points_records" 2>&1 | tail -25 | cut -c1-600
done 2>&1 | tee gpurun_out/r06/beyond_2pow32_b.txt
