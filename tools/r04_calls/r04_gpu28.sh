cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6 7 8; do for f in 1 0; do
  PST_FUSED_FOLD=$f python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-north-star 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('F=$f', d['ms_per_step'])"
done; done | python -c "
import sys
a={'F=1':[], 'F=0':[]}
for l in sys.stdin:
    k,v=l.split(); a[k].append(float(v))
for k,v in a.items(): print(k, 'n',len(v),'mean %.4f min %.4f median %.4f' % (sum(v)/len(v), min(v), sorted(v)[len(v)//2]), v)
"
