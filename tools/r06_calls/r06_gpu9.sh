cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 900 python bench.py > gpurun_out/r06/bench_line_box2.json 2> gpurun_out/r06/bench_line_box2.err ) 2>&1 | tail -3; echo rc=$?
tail -5 gpurun_out/r06/bench_line_box2.err | cut -c1-300
python - <<'P'
import json
l=json.loads(open('gpurun_out/r06/bench_line_box2.json').read().strip().splitlines()[-1])
for k in ('configs4_knn16','configs4_knn16_sheet','dropin_convert_then_bounds','chunked_rawlas_1MiB'):
    v=dict(l.get(k,{})); v.pop('note',None); print(k, json.dumps(v)[:1800]); print()
print('spot', json.dumps(l.get('cpu_baseline',{}).get('spot_checks'))[:1500])
print(l['ms_per_step'], l['roofline']['frac'], l.get('verified'))
P
