cd $GRAFT_REPO_ROOT
for nt in 7 5 3 1 0 15 13 4; do
echo "=== NT=$nt"
PST_JIT_NT=$nt timeout 900 python tools/exp_jit_layouts.py --seeds 12 --points 100000000 --steps 8 --skip-interp 2>&1 | grep -E "seed|worst|Error|error" | python -c "
import sys, json
row = {}
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); row.setdefault(d['pairing'], []).append(d['jit_frac'])
for k, v in row.items(): print(k, ' '.join('%.3f' % x for x in v), 'min %.3f mean %.3f' % (min(v), sum(v)/len(v)))
"
done
