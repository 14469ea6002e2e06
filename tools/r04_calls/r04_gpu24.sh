cd $GRAFT_REPO_ROOT
# A/B: columns -> typed LAS-0 records through the LAS transposition kernel (default) and through the plan-specialised quad kernel (PST_LAS_DECODE=0)
for rep in 1 2 3; do
  for v in 1 0; do
    PST_LAS_DECODE=$v python bench.py --no-cpu-baseline --no-north-star --workload columns_to_las0 --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PST_LAS_DECODE=$v', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"
  done
done
