cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "filter or compaction or append" 2>&1 | tail -4
for rep in 1 2 3; do for lb in 0 1; do for w in filter_big_columnar filter_big_interleaved; do
  PST_FILTER_LOOKBACK=$lb timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w lookback=$lb', d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'])"
done; done; done
