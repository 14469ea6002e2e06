cd $GRAFT_REPO_ROOT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"; }
for w in filter_las6_columnar filter_las8_columnar filter_las8_interleaved filter_las5_interleaved; do for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-north-star --workload $w --plan interpreted --steps 20 --warmup 5 2>/dev/null | tail -1 | line "$w gather"
  PST_FILTER_STREAM_MAX_BYTES=96 python bench.py --no-cpu-baseline --no-north-star --workload $w --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | line "$w stream"
done; done
