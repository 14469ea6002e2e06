cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_jit.py tests/test_filter_append.py -m gpu -x -q -k "filter or padding or tiny or compaction" 2>&1 | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"; }
for w in filter_las3_columnar filter_las3_interleaved filter_las8_columnar filter_las8_interleaved filter_las0_interleaved; do
  python bench.py --no-cpu-baseline --no-north-star --workload $w --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | line "$w"
done
