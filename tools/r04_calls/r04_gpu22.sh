cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" ) 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
