cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in libpasture_amd.so libpasture_amd_cap1024.so libpasture_amd_cap1536.so; do
  PASTURE_AMD_LIB=$PWD/pasture_amd/$lib python bench.py --workload filter_big_interleaved --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done; done
