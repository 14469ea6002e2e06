cd $GRAFT_REPO_ROOT
for wm in 8 4 16; do
  for sd in 5 12 15 20; do
    PST_JIT_WIDE_MIN=$wm timeout 600 python tools/exp_jit_layouts.py --seeds 1 --first-seed $sd --points 100000000 --steps 8 --skip-interp 2>/dev/null | grep '"pairing"' | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('WIDE_MIN=$wm seed', d['seed'], d['pairing'], d['src_record'], d['dst_record'], d['jit_frac'])"
  done
done
