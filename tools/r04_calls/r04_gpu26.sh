cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "filter" 2>&1 | tail -2
PST_FILTER_IMG_ALIGNED=1 timeout 900 python -m pytest tests -m gpu -x -q -k "filter" 2>&1 | tail -2
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"; }
for rep in 1 2 3; do
  for v in 0 1; do
    PST_FILTER_IMG_ALIGNED=$v python bench.py --no-cpu-baseline --no-north-star --workload filter_big_interleaved --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | line "interleaved ALIGNED=$v"
  done
  for v in 0 1; do
    PST_FILTER_COLS_STREAM=$v python bench.py --no-cpu-baseline --no-north-star --workload filter_big_columnar --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 | line "columnar COLS_STREAM=$v"
  done
done
