cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_jit.py tests/test_las_golden.py tests/test_las_encode.py tests/test_buffer_converter.py -m gpu -x -q 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "las or LAS or raw or full_size_1e8" 2>&1 | tail -5
for w in las0_to_columns las0_to_columns_bounds rawlas_to_records rawlas_to_columns columns_to_las0; do
  for pref in 1 0; do
    PST_LAS_PREFER_SPECIALISED=$pref python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(c['workload'].split(':')[0][:30].ljust(30), 'prefer=$pref', c.get('plan'), d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
  done
done
