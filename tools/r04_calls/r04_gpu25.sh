cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in 0 1; do
    PST_LAS_PREFER_SPECIALISED=$v python bench.py --no-cpu-baseline --no-north-star --workload columns_to_las0 --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('columns_to_las0 PREFER=$v', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"
  done
done
for v in 0 1; do PST_LAS_PREFER_SPECIALISED=$v python tools/exp_las_columns_to_records.py 2>&1 | grep -v amdgpu.ids; done
timeout 1200 python -m pytest tests -m gpu -x -q -k "las or jit or conver or static or plan" 2>&1 | tail -3
