cd $GRAFT_REPO_ROOT
for w in las0_to_columns las0_to_columns_bounds rawlas_to_columns rawlas_to_columns_bounds rawlas_to_records columns_to_las0; do
  for las in 1 0; do
    PST_LAS_DECODE=$las python bench.py --workload $w --plan specialised --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(c['workload'].split(':')[0][:30].ljust(30), 'PST_LAS_DECODE=$las', c.get('plan'), d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
  done
done
mkdir -p gpurun_out/r04
timeout 1200 python tools/exp_jit_layouts.py --seeds 24 --points 100000000 --steps 8 --out gpurun_out/r04/r04_random_layouts.jsonl 2>&1 | tail -3
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r04/r04_random_layouts.jsonl')]
for p in ('VH','HV','VV'):
    r=[x for x in rows if x['pairing']==p]
    print(p, 'n', len(r), 'jit min %.3f mean %.3f' % (min(x['jit_frac'] for x in r), sum(x['jit_frac'] for x in r)/len(r)), 'interp min %.3f mean %.3f' % (min(x['interpreted_frac'] for x in r), sum(x['interpreted_frac'] for x in r)/len(r)), 'identical', all(x['identical'] for x in r), 'jit everywhere', all('jit' in x['jit_plan'] for x in r))
    print('   below 0.72:', [(x['seed'], x['src_record'], x['dst_record'], x['jit_frac']) for x in r if x['jit_frac'] < 0.72])
PY
