cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_algorithms.py tests/test_buffer_converter.py tests/test_gpu_parity.py -m gpu -x -q -k "bounds or full_size or shard or 1e9 or convert" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
for rep in 1 2 3; do for f in 1 0; do
  PST_FUSED_FOLD=$f python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-north-star 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PST_FUSED_FOLD=$f', d['ms_per_step'], d['roofline']['kernel_ms_avg'], round(4.8/d['ms_per_step']/8,4))"
done; done
