cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for e in 1 10 1000; do
PST_BENCH_EVENT_EVERY=$e timeout 600 python bench.py --no-cpu-baseline --no-north-star --no-extra-legs --no-traffic-run --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('every $e:', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done; done
