cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_algorithms.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-north-star --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_samples'], d['roofline']['note'][-40:])"
