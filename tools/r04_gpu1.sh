cd $GRAFT_REPO_ROOT
for cfg in "64 -1 1" "64 0 1" "64 1 1" "64 -1 0" "128 -1 1" "256 -1 1" "128 0 1" "128 1 1"; do
set -- $cfg
echo "=== PST_JIT_BLK=$1 XCD=$2 NT=$3"
PST_JIT_ALIAS=0 PST_JIT_BLK=$1 PST_JIT_XCD=$2 PST_JIT_NT=$3 timeout 900 python tools/exp_jit_layouts.py --seeds 8 --points 100000000 --steps 8 --skip-interp 2>&1 | grep -E "seed|worst|Error|error" | python -c "
import sys, json
row = {}
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); row.setdefault(d['pairing'], []).append(d['jit_frac'])
    else: print(l.strip()[:60])
for k, v in row.items(): print(k, ' '.join('%.3f' % x for x in v), 'min %.3f mean %.3f' % (min(v), sum(v)/len(v)))
"
done
