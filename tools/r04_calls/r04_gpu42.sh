cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do PST_FILTER_STREAM=$v python tools/exp_filter_padded.py 2>&1 | grep -v amdgpu.ids; done; done
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"; }
for rep in 1 2; do for w in filter_las3_columnar filter_las3_interleaved; do
  python bench.py --no-cpu-baseline --no-north-star --workload $w --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | line "$w"
done; done
